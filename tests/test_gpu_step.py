"""H row of SURVEY 8a on the GPU: one fwd + CrossEntropy + bwd + Adam step of PointNet2_SSG_Clas against the committed float64
fixture (tests/golden/step_b8_n1024.npz, generator tests/golden/make_golden.py::train_step), and the Adam kernel alone against
the float64 formula (PAPC/train.py:62-65: paddle.optimizer.Adam(lr, weight_decay=float) = L2 added to the gradient)."""
import os

import numpy as np
import pytest
import torch

from papc_amd import _lib
from papc_amd.distributed import FlatAdam, FlatParams
from papc_amd.head import softmax_cross_entropy
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds
from tests.util import assert_close, copy_into_model, seeded_model_state

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bn_fed_bias(name):
    return name.endswith("bias") and (".mlp_convs." in name or name in ("fc1.bias", "fc2.bias"))


def test_train_step_matches_golden(dev):
    g = np.load(os.path.join(GOLD, "step_b8_n1024.npz"))
    B, N = 8, 1024
    model = PointNet2_SSG_Clas(num_classes=16)
    copy_into_model(model, seeded_model_state(model, int(g["weight_seed"])))
    model.drop1.p = 0.0
    model.drop2.p = 0.0                                      # the fixture was generated with dropout p = 0
    model = model.to(dev).train()
    names = [n for n, _ in model.named_parameters()]
    flat = FlatParams(model)
    opt = FlatAdam(flat, lr=1e-3, weight_decay=1e-3)
    x = torch.from_numpy(make_clouds(B, N, int(g["seed"]))).to(dev)
    s1, s2 = torch.from_numpy(g["start1"]).to(dev), torch.from_numpy(g["start2"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    flat.zero_grad()
    logits = model(x, (s1, s2))                              # train mode: the fused head (head.hip) produces the logits
    loss = softmax_cross_entropy(logits, labels)
    loss.backward()
    # nine conv layers + the head in fp32 against an all-float64 chain: the chained-model bar of the other tests
    assert_close(logits.detach().cpu().numpy(), g["logits"], 2e-4, "logits (train-mode fused head)")
    assert abs(float(loss) - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])), (float(loss), float(g["loss"]))
    params = dict(model.named_parameters())
    for n in names:
        sel = g["sel/" + n]
        got = params[n].grad.reshape(-1)[torch.from_numpy(sel).to(dev)].cpu().numpy().astype(np.float64)
        if _bn_fed_bias(n):
            # a bias feeding a train-mode BatchNorm has gradient exactly 0: both sides hold rounding noise of their own
            assert np.abs(got).max() <= 1e-4 * float(g["gmax/" + n.replace("bias", "weight")]) + 1e-12, n
            continue
        err = np.abs(got - g["grad/" + n]).max() / max(float(g["gmax/" + n]), 1e-30)
        # bar: the head (no pooling) is held to 2e-4 of max|grad| like the stack tests.  The set-abstraction gradients of this
        # WHOLE-MODEL fixture get 3e-2 (SA3 sums only 1024 rows in 8 groups: one re-routed winner is >1e-2 of a bias gradient): among ~400 000 pooled (group, channel) decisions some are within 1e-7 of a tie for every
        # weight seed (see the generator), any fp32 evaluation sends those max-pool gradients to other rows than float64, and
        # each moves ~1e-3 of a channel's gradient.  With the routing pinned the same kernels are held to 2e-4
        # (tests/test_gpu_mlp.py::test_backward_near_ties_explain_the_seed40_excess); here the bar catches wrong terms, not ulps.
        bar = max(2e-4, 3.0 * float(g["err32/" + n]), 3e-2 if n.startswith("sa") else 0.0)
        assert err <= bar, "grad %s: %.2e of max|grad| (bar %.1e, plain fp32 autograd %.1e)" % (n, err, bar, float(g["err32/" + n]))
    before = {n: params[n].detach().reshape(-1)[torch.from_numpy(g["sel/" + n]).to(dev)].cpu().numpy().astype(np.float64) for n in names}
    gpu_grad = {n: params[n].grad.reshape(-1)[torch.from_numpy(g["sel/" + n]).to(dev)].cpu().numpy().astype(np.float64) for n in names}
    opt.step(1.0)
    torch.cuda.synchronize()
    off = 0
    n_el = n_close = 0
    for n in names:
        p = params[n]
        k = p.numel()
        sel = torch.from_numpy(g["sel/" + n]).to(dev)
        m = opt.m[off:off + k][sel].cpu().numpy().astype(np.float64)
        v = opt.v[off:off + k][sel].cpu().numpy().astype(np.float64)
        new = p.detach().reshape(-1)[sel].cpu().numpy().astype(np.float64)
        off += k
        # Adam state and update recomputed in float64 from the GPU's own gradient: this isolates papc_adam_step_f32
        gd = gpu_grad[n] + 1e-3 * before[n]
        mag = np.abs(gpu_grad[n]) + 1e-3 * np.abs(before[n])       # (g and the decay term may cancel: tolerances on their magnitudes)
        assert (np.abs(m - 0.1 * gd) <= 1e-6 * 0.1 * mag + 1e-30).all(), n
        assert (np.abs(v - 0.001 * gd * gd) <= 4e-6 * 0.001 * mag * mag + 1e-38).all(), n
        want = before[n] - 1e-3 * gd / (np.abs(gd) + 1e-8)
        ok = np.abs(gd) > 1e-5 * mag                               # (where they cancel to rounding the step direction is undetermined)
        assert np.abs(new - want)[ok].max() <= 2e-7 + 1e-6 * np.abs(want).max(), n
        # ... and against the fixture's float64 step: the first Adam step is -lr * g / (|g| + eps), insensitive to the size of g
        if not _bn_fed_bias(n):
            d = np.abs(new - g["new/" + n])
            n_el += d.size
            n_close += int((d <= 2e-6).sum())
            assert d.max() <= 2.1e-3, n                       # (a gradient within rounding of 0 may step the other way)
    assert n_close >= 0.995 * n_el, (n_close, n_el)


def test_adam_kernel_vs_float64(dev):
    """papc_adam_step_f32 over three steps against the float64 formula, with a gradient scale (the 1/world of the data-parallel
    mean is folded into the kernel) and L2 weight decay."""
    lib = _lib.load()
    n = 100003
    rng = np.random.default_rng(0)
    p0 = rng.normal(size=n).astype(np.float32)
    p = torch.from_numpy(p0.copy()).to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    lr, b1, b2, eps, wd, gs = 1e-3, 0.9, 0.999, 1e-8, 1e-3, 0.25
    P, M, V = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 4):
        g = (rng.normal(size=n) * 10.0 ** rng.uniform(-6, 1, size=n)).astype(np.float32)
        _lib.check(lib.papc_adam_step_f32(p.data_ptr(), torch.from_numpy(g).to(dev).data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2,
                                          eps, wd, t, gs, _lib.stream_ptr()), "papc_adam_step_f32")
        gd = g.astype(np.float64) * gs + wd * P
        M = b1 * M + (1 - b1) * gd
        V = b2 * V + (1 - b2) * gd * gd
        P = P - lr * (M / (1 - b1 ** t)) / (np.sqrt(V / (1 - b2 ** t)) + eps)
    torch.cuda.synchronize()
    # (m is a signed sum: absolute tolerance on its scale; v a sum of squares: relative)
    assert np.allclose(m.cpu().numpy(), M, rtol=1e-5, atol=2e-7 * np.abs(M).max())
    assert np.allclose(v.cpu().numpy(), V, rtol=2e-5, atol=1e-30)
    assert np.abs(p.cpu().numpy() - P).max() <= 1e-6
