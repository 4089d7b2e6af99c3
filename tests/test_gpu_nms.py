"""GPU parity: papc_nms_f32 vs the oracle restatement of nms_gpu (bit-exact index lists)."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R

pytestmark = pytest.mark.gpu


def _boxes(rng, n, extent=200.0, size=40.0, ties=False):
    xy = rng.uniform(0, extent, (n, 2)).astype(np.float32)
    wh = rng.uniform(1, size, (n, 2)).astype(np.float32)
    sc = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    if ties:
        sc = np.round(sc * 8) / np.float32(8)
    return np.concatenate([xy, xy + wh, sc], 1).astype(np.float32)


@pytest.mark.parametrize("n,thr", [(1, 0.5), (2, 0.5), (63, 0.3), (64, 0.5), (65, 0.5), (200, 0.1), (300, 0.7)])
def test_nms_matches_literal_oracle(n, thr):
    from papc_amd.nms import nms_gpu
    rng = np.random.default_rng(n)
    dets = _boxes(rng, n, extent=80.0)
    want = [int(i) for i in R.nms_gpu(dets, thr)]
    got = nms_gpu(dets, thr)
    assert got == want


@pytest.mark.parametrize("n,thr,ties", [(1000, 0.5, False), (4097, 0.3, True), (20000, 0.5, False)])
def test_nms_large_matches_vectorised_oracle(n, thr, ties):
    from papc_amd.nms import nms_gpu_tensor
    rng = np.random.default_rng(7 + n)
    dets = _boxes(rng, n, extent=300.0 if n < 10000 else 1500.0, ties=ties)
    want = np.asarray(R.nms_vectorised(dets, thr), np.int64)
    got = nms_gpu_tensor(torch.from_numpy(dets).cuda(), thr).cpu().numpy()
    assert got.shape == want.shape and (got == want).all()
    # size-independent properties: kept boxes are pairwise below the threshold, scores descend
    sc = dets[got, 4]
    assert (np.diff(sc) <= 0).all()


def test_nms_edge_cases():
    from papc_amd.nms import nms_gpu, nms_gpu_tensor
    assert nms_gpu(np.zeros((0, 5), np.float32), 0.5) == []
    same = np.tile(np.array([[10, 10, 20, 20, 0.5]], np.float32), (130, 1))   # identical boxes, tied scores: only the last index
    assert nms_gpu(same, 0.5) == [129]
    far = np.array([[i * 100, 0, i * 100 + 10, 10, 0.1 * i] for i in range(1, 9)], np.float32)   # disjoint: all kept, score order
    assert nms_gpu(far, 0.0) == list(range(7, -1, -1))
    neg = np.array([[0, 0, 10, 10, -1.0], [0, 0, 10, 10, -2.0], [50, 50, 60, 60, 0.0]], np.float32)   # negative scores order
    assert nms_gpu(neg, 0.5) == [2, 0]
    with pytest.raises(Exception):
        nms_gpu_tensor(torch.zeros(4, 5), 0.5)
