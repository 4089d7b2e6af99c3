"""GPU parity of the "planes" path for stacks with few rows (csrc/smallm.hip, papc_amd/smallm.py): the plane-set GEMM against a
float64 product, and the whole group_all stack (sample_and_group_all + conv/BN/ReLU x L + max,
/root/reference/PAPC/models/layers/pointnet2_basic_layers.py:160-176, :215-219) against the row kernels and a float64 torch
reference -- forward 1e-5, gradients 2e-4 (the bars of tests/test_gpu_mlp.py)."""
import ctypes

import numpy as np
import pytest
import torch

from papc_amd import _lib, smallm
from papc_amd.mlp import SharedMLPMax, StackSpec, shared_mlp_max
from papc_amd.synthetic import make_clouds
from tests import torch_ref
from tests.util import assert_close, seeded_weights

pytestmark = pytest.mark.gpu


def _to_planes(lib, mats, dev):
    """[(tensor [R, K], transpose?)] -> plane buffers through papc_pg_prep_weights_f32"""
    out, arr = [], (smallm.PgWJob * len(mats))()
    for a, (m, tr) in zip(arr, mats):
        R, K = (m.shape[1], m.shape[0]) if tr else m.shape
        buf = torch.empty(lib.papc_pg_planes_bytes(R, K), dtype=torch.uint8, device=dev)
        a.src, a.R, a.K, a.planes = m.data_ptr(), R, K, buf.data_ptr()
        a.row_stride, a.col_stride = (1, m.shape[1]) if tr else (m.shape[1], 1)
        out.append(buf)
    _lib.check(lib.papc_pg_prep_weights_f32(arr, len(mats), _lib.stream_ptr()), "papc_pg_prep_weights_f32")
    return out


@pytest.mark.parametrize("R1,R2,K,split", [
    (256, 128, 64, 1),            # one tile, two stages
    (4096, 1024, 512, 1),         # SA3 layer 3 forward
    (4096, 256, 259, 1),          # ragged contraction (padded to 288), 64-wide column tiles
    (300, 259, 100, 1),           # ragged everything (masked epilogue)
    (1024, 512, 4096, 8),         # SA3 layer 3 dW: split over the contraction
    (256, 259, 4096, 16),         # layer 1 dW
])
def test_planes_gemm_vs_f64(dev, R1, R2, K, split):
    lib = _lib.load()
    rng = np.random.default_rng(R1 + R2 + K)
    # wide dynamic range: the three-plane split must carry all 24 significand bits
    a = torch.from_numpy((rng.normal(size=(R1, K)) * np.exp(rng.normal(size=(R1, K)))).astype(np.float32)).to(dev)
    b = torch.from_numpy((rng.normal(size=(R2, K)) * np.exp(rng.normal(size=(R2, K)))).astype(np.float32)).to(dev)
    pa, pb = _to_planes(lib, [(a, False), (b, False)], dev)
    c = torch.full((split, R1, R2), float("nan"), device=dev)
    g = smallm.PgGemm()
    g.epi, g.a, g.b, g.R1, g.R2, g.K = smallm.EPI_STORE, pa.data_ptr(), pb.data_ptr(), R1, R2, K
    g.c, g.ldc, g.split, g.split_stride, g.family = c.data_ptr(), R2, split, R1 * R2, 9
    _lib.check(lib.papc_pg_gemm_f32(ctypes.byref(g), _lib.stream_ptr()), "papc_pg_gemm_f32")
    ref = a.double() @ b.double().t()
    got = c.double().sum(0)
    # error bar relative to sum |a_k b_k| (what an fp32 dot product is judged by), elementwise
    mag = a.double().abs() @ b.double().abs().t()
    err = float(((got - ref).abs() / mag).max())
    assert err < 3e-6, err   # fp32 accumulation over K terms (sqrt(K) * 2^-24 typical); a lost plane or a stale fragment would be >= 1e-3
    # the transposed-source job (W^T planes straight from W)
    pbt, = _to_planes(lib, [(b.t().contiguous(), True)], dev)
    c2 = torch.empty_like(c)
    g.b, g.c = pbt.data_ptr(), c2.data_ptr()
    _lib.check(lib.papc_pg_gemm_f32(ctypes.byref(g), _lib.stream_ptr()), "papc_pg_gemm_f32")
    assert torch.equal(c, c2)
    if split > 1:   # fixed-order fold, accumulate on top of an existing gradient
        out = torch.ones(R1, R2, device=dev)
        job = (smallm.PgFoldJob * 1)()
        job[0].partial, job[0].nsplit, job[0].stride, job[0].n, job[0].out, job[0].accumulate = c.data_ptr(), split, R1 * R2, R1 * R2, out.data_ptr(), 1
        _lib.check(lib.papc_pg_fold_f32(job, 1, _lib.stream_ptr()), "papc_pg_fold_f32")
        want = torch.ones(R1, R2, device=dev)
        s = c[0].clone()
        for t in range(1, split):
            s += c[t]
        assert torch.equal(out, s + want)


def _run_stack(dev, B, D, mlp, xyz_first, plain, planes, seed, grad_targets=False):
    """one forward + backward of a group_all stack; returns (out, grads dict)"""
    N = 128
    rng = np.random.default_rng(seed)
    x = make_clouds(B, N, seed)
    xyz = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
    feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_(True)
    cin = D if plain else D + 3
    ws = seeded_weights([cin] + mlp, seed + 1)
    params = []
    for (w, b, g, bt) in ws:
        params += [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (w, b, g, bt)]
    spec = StackSpec(B, N, 1, N, D, xyz_first)
    new_xyz = torch.zeros(B, 1, 3, device=dev)
    old = smallm.ENABLED
    smallm.ENABLED = planes
    try:
        if plain:
            rows = feats.reshape(B * N, D)
            out = shared_mlp_max(spec, None, None, None, None, None, params, x_rows=rows)
        else:
            out = shared_mlp_max(spec, None, xyz, new_xyz, feats, None, params)
    finally:
        smallm.ENABLED = old
    gout = torch.from_numpy(np.random.default_rng(seed + 2).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
    out.backward(gout)
    return out.detach(), [p.grad for p in params], feats.grad, (xyz, feats, params, gout, spec)


@pytest.mark.parametrize("B,D,mlp,xyz_first,plain", [
    (32, 256, [256, 512, 1024], True, False),      # PointNet2_SSG_Clas.sa3 at the BASELINE batch (M = 4096)
    (4, 640, [256, 512, 1024], True, False),       # PointNet2_MSG_Clas.sa3 input width
    (2, 64, [64, 128], True, False),               # two layers, one 64-column tile
    (3, 40, [48, 72, 24], False, False),           # ragged widths (masked epilogues), feats-first rows
    (8, 128, [128, 256], True, True),              # plain rows (no concat)
])
def test_planes_stack_vs_rows_and_f64(dev, B, D, mlp, xyz_first, plain):
    seed = 40 + B
    out, grads, gfeat, (xyz, feats, params, gout, spec) = _run_stack(dev, B, D, mlp, xyz_first, plain, True, seed)
    node = out.grad_fn
    out_r, grads_r, gfeat_r, _ = _run_stack(dev, B, D, mlp, xyz_first, plain, False, seed)
    # (a) against the row kernels: same function, other summation order
    assert_close(out.cpu().numpy(), out_r.cpu().numpy(), 2e-6, "planes vs rows forward")
    # (b) against float64 torch
    N = 128
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    f64 = feats.detach().double().requires_grad_(True)
    if plain:
        rows = f64.reshape(B * N, D)
    else:
        ridx = torch.arange(N, device=dev).view(1, 1, N).expand(B, 1, N)
        rows = torch_ref.group(xyz.double(), torch.zeros(B, 1, 3, device=dev).double(), f64, ridx, xyz_first).reshape(B * N, D + 3)
    ref = torch_ref.stack_max(rows, [tuple(p64[4 * l:4 * l + 4]) for l in range(len(mlp))], N, 1e-5)
    assert_close(out.cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "planes forward vs f64")
    ref.backward(gout.double())
    names = ["w", "b", "gamma", "beta"]
    for l in range(len(mlp)):
        for j in range(4):
            got, want = grads[4 * l + j], p64[4 * l + j].grad
            if j == 1:
                assert float(got.abs().max()) == 0.0          # conv bias under a train-mode BN: exact zero
                continue
            assert_close(got.cpu().numpy(), want.cpu().numpy(), 2e-4, "planes d%s layer %d" % (names[j], l))
            assert_close(got.cpu().numpy(), grads_r[4 * l + j].cpu().numpy(), 2e-4, "planes vs rows d%s layer %d" % (names[j], l))
    assert_close(gfeat.cpu().numpy(), f64.grad.cpu().numpy(), 2e-4, "planes dfeats")
    assert_close(gfeat.cpu().numpy(), gfeat_r.cpu().numpy(), 2e-4, "planes vs rows dfeats")


def test_planes_path_is_taken_and_deterministic(dev):
    """the group_all layer really runs on smallm.hip (not silently on the row kernels), bit-identical from run to run, and honours
    the in-place gradient targets"""
    B, D, mlp = 8, 256, [256, 512, 1024]
    out1, grads1, gf1, (xyz, feats, params, gout, spec) = _run_stack(dev, B, D, mlp, True, False, True, 7)
    out2, grads2, gf2, _ = _run_stack(dev, B, D, mlp, True, False, True, 7)
    assert torch.equal(out1, out2) and torch.equal(gf1, gf2)
    for a, b in zip(grads1, grads2):
        assert torch.equal(a, b)
    spec2 = StackSpec(B, 128, 1, 128, D, True)
    assert smallm.eligible(spec2, xyz, feats.detach(), None, None, params)
    o = shared_mlp_max(spec2, None, xyz, torch.zeros(B, 1, 3, device=dev), feats.detach(), None, params)
    assert getattr(o.grad_fn, "planes", False) or "PlanesMLPMax" in type(o.grad_fn).__name__, type(o.grad_fn).__name__
    # in-place targets: gradients are ADDED into the given buffers
    tg = [torch.ones_like(p) for p in params]
    spec3 = StackSpec(B, 128, 1, 128, D, True)
    spec3.grad_targets = tg
    from papc_amd.stack import SharedMLPStack
    o3 = SharedMLPStack.apply(spec3, None, xyz, torch.zeros(B, 1, 3, device=dev), feats.detach(), None, None, *params)
    assert o3.grad_fn.planes
    for p in params:
        p.grad = None
    o3.backward(gout)
    for l in range(len(mlp)):
        for j in (0, 2, 3):
            assert params[4 * l + j].grad is None
            assert torch.allclose(tg[4 * l + j] - 1.0, grads1[4 * l + j].reshape(tg[4 * l + j].shape), rtol=1e-4, atol=3e-6), (l, j)


@pytest.mark.parametrize("M,chans", [(2048, [1536, 256, 256]), (8192, [576, 256, 128]), (1024, [64, 128, 64])])
def test_planes_pointwise_stack_vs_rows_and_f64(dev, M, chans):
    """The planes path on a stack WITHOUT pooling (the point-wise conv/BN/ReLU stacks of feature propagation,
    /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:331-333, at the part-segmentation models' fp3 / fp2 sizes): against the
    row kernels and float64 torch, forward 1e-5, gradients (weights, norms, input rows) 2e-4."""
    rng = np.random.default_rng(M)
    x0 = torch.from_numpy(rng.normal(size=(M, chans[0])).astype(np.float32)).to(dev)
    ws = seeded_weights(chans, 43)
    gout = torch.from_numpy(rng.normal(size=(M, chans[-1])).astype(np.float32)).to(dev)

    def run(planes):
        x = x0.clone().requires_grad_(True)
        ps = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        spec = StackSpec(1, M, M, 1, chans[0], True, pool=False)
        old = smallm.ENABLED
        smallm.ENABLED = planes
        try:
            out = shared_mlp_max(spec, None, None, None, None, None, ps, x_rows=x)
        finally:
            smallm.ENABLED = old
        assert bool(getattr(out.grad_fn, "planes", False) or "Planes" in type(out.grad_fn).__name__) == planes
        out.backward(gout)
        return out.detach(), [p.grad for p in ps], x.grad

    out_p, g_p, gx_p = run(True)
    out_r, g_r, gx_r = run(False)
    assert_close(out_p.cpu().numpy(), out_r.cpu().numpy(), 2e-6, "planes vs rows forward (no pooling)")
    p64 = [torch.from_numpy(a).to(dev).double().requires_grad_(True) for tup in ws for a in tup]
    x64 = x0.double().requires_grad_(True)
    h = x64
    for l in range(len(chans) - 1):
        w, b, g, be = p64[4 * l:4 * l + 4]
        y = h @ w.reshape(w.shape[0], -1).t() + b
        mu, var = y.mean(0), y.var(0, unbiased=False)
        h = torch.relu((y - mu) / torch.sqrt(var + 1e-5) * g + be)
    assert_close(out_p.cpu().numpy(), h.detach().cpu().numpy(), 1e-5, "planes forward vs f64 (no pooling)")
    h.backward(gout.double())
    for l in range(len(chans) - 1):
        for j, nm in ((0, "w"), (2, "gamma"), (3, "beta")):
            want = p64[4 * l + j].grad.cpu().numpy()
            assert_close(g_p[4 * l + j].cpu().numpy().reshape(want.shape), want, 2e-4, "planes d%s layer %d vs f64" % (nm, l))
            assert_close(g_p[4 * l + j].cpu().numpy(), g_r[4 * l + j].cpu().numpy(), 2e-4, "planes vs rows d%s layer %d" % (nm, l))
    assert_close(gx_p.cpu().numpy(), x64.grad.cpu().numpy(), 2e-4, "planes dx vs f64")
    assert_close(gx_p.cpu().numpy(), gx_r.cpu().numpy(), 2e-4, "planes vs rows dx")
