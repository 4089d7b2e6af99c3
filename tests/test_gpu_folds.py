"""Deferred folds (papc_amd/folds.py, papc_fold_jobs_f32): the partial reductions of a whole backward pass in one launch.

(i) the kernel against numpy on both of its shapes (many chunks / few chunks), strided outputs, accumulation;
(ii) a training step of PointNet2_SSG_Clas (reference: PAPC/models/classify/pointnet2/pointnet2.py:6-41, loop PAPC/train.py:106-116)
with the folds deferred against the same step with every stack folding its own partials: identical losses, gradients equal up to
the summation order of the strided / split-K jobs (the dW / db jobs keep their order and are bit-identical);
(iii) the same inside a captured hipGraph."""
import ctypes

import numpy as np
import pytest
import torch

from papc_amd import _lib, folds
from papc_amd.distributed import FlatParams
from papc_amd.head import softmax_cross_entropy
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds, make_labels, make_start_idx
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def test_fold_jobs_kernel_vs_numpy(dev):
    lib = _lib.load()
    rng = np.random.default_rng(5)
    specs = [  # (n_chunks, rows, cols, out_ld, ld pad, accumulate)
        (200, 1, 64 * 3, 64 * 3, 0, 0), (37, 128, 3, 131, 5, 1), (512, 1, 77, 77, 0, 1), (8, 1, 1024 * 512, 1024 * 512, 0, 1),
        (3, 1, 16384 + 4, 16384 + 4, 0, 0), (1, 1, 20000, 20000, 0, 0), (64, 64, 128, 200, 64, 0), (16, 1, 256 * 128, 256 * 128, 128, 1)]
    jobs = (folds.FoldJob * len(specs))()
    keep, refs = [], []
    for i, (nc, rows, cols, out_ld, pad, acc) in enumerate(specs):
        ld = rows * cols + pad
        part = rng.normal(size=(nc, ld)).astype(np.float32)
        out0 = rng.normal(size=(rows, out_ld)).astype(np.float32)
        tp, to = torch.from_numpy(part).to(dev), torch.from_numpy(out0).to(dev)
        keep += [tp, to]
        ref = out0.astype(np.float64).copy()
        blk = part[:, :rows * cols].astype(np.float64).sum(0).reshape(rows, cols)
        ref[:, :cols] = (ref[:, :cols] if acc else 0.0) + blk
        refs.append((to, ref, cols))
        jobs[i] = folds.FoldJob(tp.data_ptr(), nc, acc, ld, rows, cols, to.data_ptr(), out_ld)
    _lib.check(lib.papc_fold_jobs_f32(jobs, len(specs), _lib.stream_ptr()), "papc_fold_jobs_f32")
    torch.cuda.synchronize()
    for i, (to, ref, cols) in enumerate(refs):
        got = to.cpu().numpy()
        assert_close(got[:, :cols], ref[:, :cols], 2e-6, "fold job %d" % i)
        assert np.array_equal(got[:, cols:], ref[:, cols:].astype(np.float32)), "job %d wrote outside its block" % i


def _step(dev, defer, graph):
    old = folds.ENABLED
    folds.ENABLED = defer
    try:
        B, N = 4, 1024
        torch.manual_seed(11)
        model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
        model.drop1.p = model.drop2.p = 0.0
        flat = FlatParams(model)
        x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
        y = torch.from_numpy(make_labels(B, 16, 3)).reshape(-1).to(dev)
        st = (torch.from_numpy(make_start_idx(B, N, 3)).to(dev), torch.from_numpy(make_start_idx(B, 512, 4)).to(dev))
        one = torch.ones((), device=dev)

        def fwd_bwd():
            loss = softmax_cross_entropy(model(x, st), y)
            loss.backward(one)
            return loss

        if graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fwd_bwd()
                flat.zero_grad()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    loss = fwd_bwd()
                flat.zero_grad()
                g.replay()
            torch.cuda.current_stream().wait_stream(s)
        else:
            flat.zero_grad()
            loss = fwd_bwd()
        torch.cuda.synchronize()
        return float(loss), flat.grad.detach().cpu().numpy().copy()
    finally:
        folds.ENABLED = old


def test_deferred_folds_equal_per_stack_folds(dev):
    l0, g0 = _step(dev, False, False)
    for graph in (False, True):
        l1, g1 = _step(dev, True, graph)
        assert l0 == l1, (l0, l1)
        # (the gather-add backward's float atomics make two runs differ in the last bits whatever the fold does)
        assert_close(g1, g0, 2e-5, "flat gradient, deferred vs per-stack folds (graph=%s)" % graph)
    assert np.abs(g0).max() > 0


def test_failed_backward_leaves_no_stale_fold_jobs(dev):
    """A backward that raises after a stack has queued its folds (here: a hook on l2_points, i.e. behind SA3's backward) never runs the
    engine's final callbacks.  The NEXT pass must fold exactly its own jobs: gradients equal the per-stack-fold result (round-5 advisor
    finding: the armed thread-local list kept the stale jobs and skipped the registration of every later pass)."""
    l0, g0 = _step(dev, False, False)
    old = folds.ENABLED
    folds.ENABLED = True
    try:
        B, N = 4, 1024
        torch.manual_seed(11)
        model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
        model.drop1.p = model.drop2.p = 0.0
        flat = FlatParams(model)
        x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
        y = torch.from_numpy(make_labels(B, 16, 3)).reshape(-1).to(dev)
        st = (torch.from_numpy(make_start_idx(B, N, 3)).to(dev), torch.from_numpy(make_start_idx(B, 512, 4)).to(dev))
        one = torch.ones((), device=dev)

        def boom(g):
            raise RuntimeError("injected failure behind SA3's backward")

        tap = {}
        loss = softmax_cross_entropy(model(x, st, tap=tap), y)
        tap["l2_points"].register_hook(boom)
        with pytest.raises(RuntimeError, match="injected"):
            loss.backward(one)
        torch.cuda.synchronize()
        assert len(folds._live) == 1 and next(iter(folds._live.values())).lst.count > 0     # the failed pass's jobs were never folded
        flat.zero_grad()
        loss = softmax_cross_entropy(model(x, st), y)
        loss.backward(one)
        torch.cuda.synchronize()
        assert float(loss) == l0
        assert_close(flat.grad.detach().cpu().numpy(), g0, 2e-5, "flat gradient of the pass after a failed one")
    finally:
        folds.ENABLED = old
        folds._live.clear()
