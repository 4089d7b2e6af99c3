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
    # the model's own forward, level by level (what PointNet2_SSG_Clas.forward does with a plan), so that the kernels' max-pool decisions can be
    # read from the stacks' autograd nodes for the routed check below
    from tests.util import kernel_decisions
    plan = model.plan_sampling(x, (s1, s2))
    l1_xyz, l1 = model.sa1(x, None, s1, sampled=plan[0])
    l2_xyz, l2 = model.sa2(l1_xyz, l1, s2, sampled=plan[1])
    _, l3 = model.sa3(l2_xyz, l2)
    logits = model._head(l3.reshape(B, 1024))                # train mode: the fused head (head.hip) produces the logits
    with torch.no_grad():
        assert torch.equal(logits, model(x, (s1, s2))), "model.forward and the level-by-level chain differ"
    argmaxes = [kernel_decisions(t)[0].clone() for t in (l1, l2, l3)]
    pooled = [t.detach().transpose(1, 2).reshape(-1, t.shape[1]) for t in (l1, l2, l3)]
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
    # ... and what that 3e-2 allowance would let through is closed here, on the fixture's own inputs and weights: the float64 chain ROUTED through
    # the kernels' decisions (round-5 review: "this fixture would not notice a 1 % wrong term in SA1 / SA2 dW").  (i) the routed chain IS the
    # fixture's computation up to routing: its logits and loss equal the committed ones to 1e-9 (the forward is continuous in the routing); (ii)
    # only a handful of alive / dead decisions differ from float64's own; (iii) every SA gradient of the kernels is within 2e-4 of the routed chain's (3x plain
    # fp32 autograd on the same routed graph where that is worse) -- all elements, not the fixture's sample
    stats = {}
    p64, lg64, ls64 = _routed_reference(model, x, plan, labels, argmaxes, pooled, torch.float64, stats)
    p32, _, _ = _routed_reference(model, x, plan, labels, argmaxes, pooled, torch.float32, stats)
    assert_close(lg64.cpu().numpy(), g["logits"], 1e-7, "routed float64 chain vs the fixture's logits", elem=1e-5)
    assert abs(ls64 - float(g["loss"])) <= 1e-7 * abs(float(g["loss"]))
    for nm, (moved, flips, ndec) in stats.items():
        print("%s: %d winner rows and %d alive/dead decisions differ from float64's own (of %d)" % (nm, moved, flips, ndec))
        # (winner rows: ball-query padding copies are exact ties, and torch's argmax does not promise the first of them -- only the alive / dead
        # decisions are held to "a handful")
        assert flips <= max(4, 2e-4 * ndec), (nm, moved, flips, ndec)
    bad = []
    for n in names:
        if _bn_fed_bias(n):
            continue
        want = p64[n].grad
        scale = float(want.abs().max())
        ours = float((params[n].grad.double() - want).abs().max()) / scale
        e32 = float((p32[n].grad.double() - want).abs().max()) / scale
        if ours > max(2e-4, 3.0 * e32):
            bad.append("%s: %.2e of max|grad| (plain fp32 autograd %.2e)" % (n, ours, e32))
    assert not bad, bad
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

def test_adam_step_zero_equals_step_then_clear(dev):
    """papc_adam_step_zero_f32 == papc_adam_step_f32 followed by clearing the gradient: bit-identical parameters and moments, gradient all zeros"""
    lib = _lib.load()
    n = 100003
    rng = np.random.default_rng(5)
    p0 = torch.from_numpy(rng.normal(size=n).astype(np.float32)).to(dev)
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    for t in range(1, 4):
        g = torch.from_numpy((rng.normal(size=n) * 10.0 ** rng.uniform(-6, 1, size=n)).astype(np.float32)).to(dev)
        ga, gb = g.clone(), g.clone()
        _lib.check(lib.papc_adam_step_f32(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, t, 0.5, _lib.stream_ptr()), "adam")
        _lib.check(lib.papc_adam_step_zero_f32(pb.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, t, 0.5, _lib.stream_ptr()), "adam")
        torch.cuda.synchronize()
        assert torch.equal(ga, g) and not gb.any()
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)



def _stack_node(t):
    """the autograd node of the MLP stack behind a set-abstraction output [B, D', S] (transpose <- view <- stack)"""
    fn = t.grad_fn
    for _ in range(4):
        if "MLPMax" in type(fn).__name__ or "MLPStack" in type(fn).__name__:
            return fn
        fn = fn.next_functions[0][0]
    raise AssertionError("no stack node behind " + type(t.grad_fn).__name__)


def _routed_reference(model, x, plan, labels, argmaxes, pooled, dt, stats):
    """One step of PointNet2_SSG_Clas (sampling plan given -> SA1 -> SA2 -> SA3 -> FC head -> cross-entropy; classify/pointnet2/pointnet2.py:33-39,
    train.py:106-109) as plain torch autograd in dtype ``dt``, ROUTED through the kernels' own max-pool decisions: winner row of every (group,
    channel) = ``argmaxes[level]``, alive iff the kernel's pooled output ``pooled[level]`` is > 0.  Every arithmetic term is the reference's own;
    only the discontinuous choices are pinned.  Returns ({name: leaf with .grad}, logits, loss); ``stats[level]`` = (winner rows, alive/dead
    decisions that differ from float64's own, decisions)."""
    from tests import torch_ref
    dev, B = x.device, x.shape[0]
    ps = {n: p.detach().to(dt).requires_grad_(True) for n, p in model.named_parameters()}
    xyz = x.transpose(1, 2).to(dt)
    feats = None
    cur_xyz = xyz
    levels = [("sa1", plan[0], 32, True), ("sa2", plan[1], 64, True), ("sa3", None, 128, True)]
    for li, (nm, pl, K, xyz_first) in enumerate(levels):
        if pl is not None:
            new_xyz, idx = pl[0].to(dt), pl[1]
            S = new_xyz.shape[1]
        else:
            S = 1
            new_xyz = torch.zeros(B, 1, 3, device=dev, dtype=dt)
            idx = torch.arange(cur_xyz.shape[1], device=dev).view(1, 1, -1).expand(B, 1, -1)
        act = torch_ref.group(cur_xyz, new_xyz, feats, idx, xyz_first).reshape(B * S * K, -1)
        z = None
        for l in range(3):
            w = ps["%s.mlp_convs.%d.weight" % (nm, l)].reshape(-1, act.shape[1])
            y = act @ w.t() + ps["%s.mlp_convs.%d.bias" % (nm, l)]
            z = (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5) * ps["%s.mlp_bns.%d.weight" % (nm, l)] + ps["%s.mlp_bns.%d.bias" % (nm, l)]
            act = torch.relu(z)
        C = z.shape[1]
        z = z.reshape(B * S, K, C)
        route = argmaxes[li].long().unsqueeze(1)
        zr = z.gather(1, route).squeeze(1)
        alive = pooled[li] > 0
        if dt == torch.float64:
            tm = torch.relu(z).max(1).values.detach()
            tol = 1e-5 * float(tm.abs().max())
            assert bool((torch.relu(zr.detach()) >= tm - tol).all()), "%s: a kernel winner is not within 1e-5 of the max" % nm
            stats[nm] = (int((route.squeeze(1) != torch.relu(z).argmax(1)).sum()), int((alive != (zr > 0)).sum()), alive.numel())
        feats = torch.where(alive, zr, torch.zeros_like(zr)).reshape(B, S, C)
        cur_xyz = new_xyz
    h = feats.reshape(B, 1024)
    for i, (fc, bn) in enumerate((("fc1", "bn1"), ("fc2", "bn2"))):
        y = h @ ps[fc + ".weight"].t() + ps[fc + ".bias"]
        h = torch.relu((y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5) * ps[bn + ".weight"] + ps[bn + ".bias"])
    lg = h @ ps["fc3.weight"].t() + ps["fc3.bias"]
    ls = torch.nn.functional.cross_entropy(lg, labels)
    ls.backward()
    return ps, lg.detach(), float(ls.detach())


@pytest.mark.parametrize("B", [4, 8])     # B = 8: SA1 (M = 131072) and SA2 (M = 65536) run the kernels the bench times -- row-streaming forward / dX, dw_rows / dw_rowsx,
def test_whole_model_gradients_with_pinned_routing(dev, B):   # the gather-add first layer, SA1's moment path and no-store max layer -- not the tiled fallback of B = 4
    """Every parameter gradient of one PointNet2_SSG_Clas step (sampling -> SA1 -> SA2 -> SA3 -> FC head -> cross-entropy,
    classify/pointnet2/pointnet2.py:33-39, train.py:106-109) against float64 torch autograd of the same graph at 2e-4 of max |grad|.
    A max-pooled gradient is a discontinuous function of the activations, so the float64 reference is routed through the kernels' own
    decisions (winner row of every (group, channel); alive iff the kernel's pooled output is > 0) -- with routing equal a missing or
    mis-scaled term anywhere in the SA2 / SA1 backward shows up at its full size instead of hiding under the golden fixture's 3e-2
    near-tie allowance.  Where plain fp32 torch autograd on the same routed graph is itself worse than 2e-4 / 3 (ill-conditioned
    first-layer sums), the bar is 3x that.  Reports how many decisions differ from float64's own."""
    from tests import torch_ref
    N = 1024
    model = PointNet2_SSG_Clas(num_classes=16)
    copy_into_model(model, seeded_model_state(model, 99))
    model.drop1.p = model.drop2.p = 0.0
    model = model.to(dev).train()
    x = torch.from_numpy(make_clouds(B, N, 7)).to(dev)
    from papc_amd.synthetic import make_labels, make_start_idx
    s1, s2 = torch.from_numpy(make_start_idx(B, N, 7)).to(dev), torch.from_numpy(make_start_idx(B, 512, 8)).to(dev)
    labels = torch.from_numpy(make_labels(B, 16, 7)).reshape(-1).to(dev)
    plan = model.plan_sampling(x, (s1, s2))
    l1_xyz, l1 = model.sa1(x, None, s1, sampled=plan[0])
    l2_xyz, l2 = model.sa2(l1_xyz, l1, s2, sampled=plan[1])
    l3_xyz, l3 = model.sa3(l2_xyz, l2)
    logits = model._head(l3.reshape(B, 1024))
    loss = softmax_cross_entropy(logits, labels)
    nodes = [_stack_node(t) for t in (l1, l2, l3)]
    assert getattr(nodes[2], "planes", False) or "Planes" in type(nodes[2]).__name__      # the group_all layer runs on the planes kernels
    from tests.util import kernel_decisions
    argmaxes = [kernel_decisions(t)[0].clone() for t in (l1, l2, l3)]      # (winner offsets in the padded groups, also for a compacted stack)
    print("stack kernels:", [type(n).__name__ + (" (compacted)" if getattr(n, "compact", None) is not None else "") for n in nodes])
    pooled = [l1.detach().transpose(1, 2).reshape(-1, l1.shape[1]), l2.detach().transpose(1, 2).reshape(-1, l2.shape[1]),
              l3.detach().transpose(1, 2).reshape(-1, l3.shape[1])]
    loss.backward()
    stats = {}

    p64, lg64, ls64 = _routed_reference(model, x, plan, labels, argmaxes, pooled, torch.float64, stats)
    p32, _, _ = _routed_reference(model, x, plan, labels, argmaxes, pooled, torch.float32, stats)
    assert_close(logits.detach().cpu().numpy(), lg64.cpu().numpy(), 2e-4, "logits vs routed f64 chain")
    assert abs(float(loss) - ls64) <= 2e-4 * abs(ls64)
    for nm, (moved, flips, n) in stats.items():
        print("%s: %d winner rows and %d alive/dead decisions differ from float64's own (of %d)" % (nm, moved, flips, n))
    bad = []
    for n, p in model.named_parameters():
        want = p64[n].grad
        if _bn_fed_bias(n):
            continue
        scale = float(want.abs().max())
        ours = float((p.grad.double() - want).abs().max()) / scale
        e32 = float((p32[n].grad.double() - want).abs().max()) / scale
        if ours > max(2e-4, 3.0 * e32):
            bad.append("%s: %.2e of max|grad| (plain fp32 autograd %.2e)" % (n, ours, e32))
    assert not bad, bad


def test_adam_device_step_count_equals_host_form(dev):
    """FlatAdam.step_dev (papc_adam_step_dev_f32: step count in device memory, advanced by papc_adam_tick -- the graph-capturable form)
    against FlatAdam.step (host scalar) over several steps on identical gradients, eagerly and replayed from a captured hipGraph.
    PAPC/train.py:62-65."""
    import torch.nn as nn
    torch.manual_seed(3)
    gens = [torch.randn(7, 1536, device=dev) for _ in range(2)]

    def run(mode):
        torch.manual_seed(5)
        mod = nn.Linear(1536, 1, bias=False).to(dev)
        flat = FlatParams(mod)
        opt = FlatAdam(flat, lr=1e-3, weight_decay=1e-3)
        if mode == "graph":
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    opt.tick()
                    opt.step_dev(0.5, zero_grad=True)
            torch.cuda.current_stream().wait_stream(s)
        if mode == "graph_selftick":
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    opt.step_dev(0.5, zero_grad=True, self_tick=True)
            torch.cuda.current_stream().wait_stream(s)
        for k in range(7):
            flat.grad.copy_(gens[0][k] if k % 2 == 0 else gens[1][k])
            if mode == "host":
                opt.step(0.5, zero_grad=True)
            elif mode == "dev":
                opt.tick()
                opt.step_dev(0.5, zero_grad=True)
            elif mode == "selftick":         # the launch advances the count itself (its last-finishing block; 6 blocks here)
                opt.step_dev(0.5, zero_grad=True, self_tick=True)
            elif mode == "mixed":            # eager warm-up on the host count, then device-counted steps, then the host path again: ONE count
                if k < 2 or k == 6:
                    opt.step(0.5, zero_grad=True)
                else:
                    opt.tick()
                    opt.step_dev(0.5, zero_grad=True)
            else:
                g.replay()
        torch.cuda.synchronize()
        assert float(flat.grad.abs().max()) == 0.0
        assert opt.step_count() == 7
        if mode not in ("host", "mixed"):
            assert int(opt.t_dev[0].item()) == 7 and int(opt.t_dev[1].item()) == 0
        sd = opt.state_dict()
        assert sd["step"] == 7
        opt2 = FlatAdam(flat)
        opt2.load_state_dict(sd)
        assert opt2.step_count() == 7 and torch.equal(opt2.m, opt.m)
        return flat.data.detach().cpu().numpy().copy(), opt.m.cpu().numpy().copy(), opt.v.cpu().numpy().copy()

    ref = run("host")
    for mode in ("dev", "graph", "selftick", "graph_selftick", "mixed"):
        got = run(mode)
        for a, b, nm in zip(got, ref, ("param", "exp_avg", "exp_avg_sq")):
            assert_close(a, b, 1e-6, "Adam %s, %s step count vs host scalar" % (nm, mode))


def test_unit_gradient_seed_skips_the_scale_launch(dev):
    """softmax_cross_entropy's backward seeded with head.unit_gradient() hands the stored gradient on as it is; any other seed is multiplied in"""
    from papc_amd.head import unit_gradient
    z = torch.randn(8, 16, device=dev, requires_grad=True)
    y = torch.randint(0, 16, (8,), device=dev)
    softmax_cross_entropy(z, y).backward(unit_gradient(dev))
    g1 = z.grad.clone()
    z.grad = None
    softmax_cross_entropy(z, y).backward()
    assert torch.equal(g1, z.grad)
    z.grad = None
    softmax_cross_entropy(z, y).backward(torch.full((), 2.0, device=dev))
    assert torch.equal(2.0 * g1, z.grad)


def test_flag_gate_orders_two_streams(dev):
    """papc_flag_wait on one stream holds that stream's later launches back until papc_flag_set runs on another (the device-side gate of
    bench.py's side-graph structure).  The gate counts: word 0 = openings, word 1 = openings waited for, word 2 = sticky count of waits that gave
    up; the opening launch also advances an int64 counter.  A wait that gives up and the LATE opening behind it must not let the next wait
    through early (round 5's single word did: it was returned to zero by the wait, so the late store left a stale 1)."""
    lib = _lib.load()
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    buf = torch.zeros(1 << 20, device=dev)
    out = torch.empty_like(buf)
    torch.cuda.synchronize()
    for rep in range(5):
        # (the setter is enqueued first, as the header asks: two streams may share a hardware queue, where a gate ahead of its setter would spin
        # until it gives up)
        with torch.cuda.stream(a):
            torch.cuda._sleep(2_000_000)                 # ~1 ms of spinning on stream a
            buf.fill_(float(rep + 1))
            _lib.check(lib.papc_flag_set(flag.data_ptr(), 1, cnt.data_ptr(), a.cuda_stream), "papc_flag_set")
        with torch.cuda.stream(b):                       # its copy must see what stream a wrote before opening the gate
            _lib.check(lib.papc_flag_wait(flag.data_ptr(), 400000, b.cuda_stream), "papc_flag_wait")
            out.copy_(buf)
        torch.cuda.synchronize()
        assert float(out.min()) == float(out.max()) == float(rep + 1)
        assert flag.tolist() == [rep + 1, rep + 1, 0, 0] and int(cnt[0]) == rep + 1
    # a wait nobody opens in time: bounded spinning, counted, and the late opening does NOT open the next wait
    with torch.cuda.stream(b):
        _lib.check(lib.papc_flag_wait(flag.data_ptr(), 200, b.cuda_stream), "papc_flag_wait")
    torch.cuda.synchronize()
    assert flag.tolist() == [5, 6, 1, 0]
    with torch.cuda.stream(a):                           # the late opening (number 6)
        _lib.check(lib.papc_flag_set(flag.data_ptr(), 1, None, a.cuda_stream), "papc_flag_set")
    torch.cuda.synchronize()
    with torch.cuda.stream(a):                           # opening number 7 arrives ~1 ms after the wait for it starts
        torch.cuda._sleep(2_000_000)
        buf.fill_(77.0)
        _lib.check(lib.papc_flag_set(flag.data_ptr(), 1, None, a.cuda_stream), "papc_flag_set")
    with torch.cuda.stream(b):
        _lib.check(lib.papc_flag_wait(flag.data_ptr(), 400000, b.cuda_stream), "papc_flag_wait")
        out.copy_(buf)
    torch.cuda.synchronize()
    assert float(out.min()) == float(out.max()) == 77.0, "the wait behind a late opening passed one opening early"
    assert flag.tolist() == [7, 7, 1, 0]
