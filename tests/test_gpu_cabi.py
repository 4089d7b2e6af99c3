"""The coarse C entry points of the MLP half of the boundary -- papc_sa_mlp_plan / _fwd / _bwd (include/papc_hip.h, csrc/sa_mlp.hip), the
one-call-per-direction replacement of the body of PointNetSetAbstraction.forward (pointnet2_basic_layers.py:214-219) -- driven

  * through RAW ctypes with nothing of papc_amd.mlp / papc_amd.stack in between (torch only owns the device memory): SA1-, SA2- and
    SA3-shaped stacks of BASELINE configs[1] against the float64 oracle, forward 1e-5, gradients 2e-4 (routed through the kernels' own
    max / ReLU decisions, read from the `saved` buffer at the offsets the plan reports);
  * against the launch sequence spelled out in Python (papc_amd.mlp.SharedMLPMax, PAPC_PY_ORCH=1): the same kernels in the same order,
    so outputs and gradients must agree BIT FOR BIT (float atomics of the gather-add backward excepted)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import reference_np as R
from papc_amd import _lib
from papc_amd import compact as C
from papc_amd import functional as F
from papc_amd import mlp as M_
from papc_amd.mlp import StackSpec
from papc_amd.stack import CompactSrc, SaDesc, SaGrads, SaIo, SaPlan, SharedMLPStack
from papc_amd.synthetic import make_clouds, make_start_idx
from tests import torch_ref
from tests.util import assert_close, seeded_weights

pytestmark = pytest.mark.gpu


def _sample(dev, B, N, S, K, radius, seed):
    x = make_clouds(B, N, seed)
    xyz = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
    st = torch.from_numpy(make_start_idx(B, N, seed)).to(dev)
    _, new_xyz = F._fps_raw(xyz, S, st)
    idx = F._ball_query_raw([radius], [K], xyz, new_xyz)[0]
    return xyz, new_xyz, idx


def _raw_stack(dev, B, N, S, K, D, chans, xyz, new_xyz, feats, idx, ws, gout, compact=None, want_feats_grad=False):
    """one forward + backward of a stack through the three C entry points only; returns (out, grads per layer, grad_feats, plan, views)"""
    lib = _lib.load()
    L = len(chans) - 1
    st = torch.cuda.current_stream().cuda_stream
    par = [[torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in tup] for tup in ws]
    d = SaDesc()
    d.B, d.N, d.S, d.K, d.D, d.n_layers, d.cin = B, N, S, K, D, L, D + 3
    for l in range(L):
        d.cout[l] = chans[l + 1]
    d.input, d.identity_rows, d.xyz_first, d.pool, d.eval_bn = 0, int(idx is None), 1, 1, 0
    d.eps, d.momentum = 1e-5, 0.9
    io = SaIo()
    io.xyz, io.sb, io.sn, io.sc = xyz.data_ptr(), xyz.stride(0), xyz.stride(1), xyz.stride(2)
    io.new_xyz = new_xyz.data_ptr()
    io.feats = feats.data_ptr() if feats is not None else None
    io.idx = idx.data_ptr() if idx is not None else None
    keep = []
    if compact is not None:
        src = CompactSrc(compact.start.data_ptr(), compact.rows.data_ptr(), compact.cidx.data_ptr(), compact.seg_grp.data_ptr(),
                         compact.wrow.data_ptr(), compact.coef.data_ptr(), compact.G)
        keep.append(src)
        io.compact = ctypes.pointer(src)
    c3 = torch.tensor([1.0, 0.0, 1e30], device=dev).view(3, 1).expand(3, 1024).contiguous()
    io.consts3, io.consts3_ld = c3.data_ptr(), 1024
    for l in range(L):
        io.layer[l].w, io.layer[l].b, io.layer[l].gamma, io.layer[l].beta = (t.data_ptr() for t in par[l])
    plan = SaPlan()
    _lib.check(lib.papc_sa_mlp_plan(ctypes.byref(d), ctypes.byref(io), ctypes.byref(plan)), "papc_sa_mlp_plan")
    G, Mrows = B * S, B * S * K
    out = torch.empty(G, chans[-1], device=dev)
    saved = torch.empty(plan.saved_bytes, device=dev, dtype=torch.uint8)
    scr = torch.empty(max(plan.fwd_scratch_bytes, plan.bwd_scratch_bytes), device=dev, dtype=torch.uint8)
    io.out, io.saved, io.scratch = out.data_ptr(), saved.data_ptr(), scr.data_ptr()
    _lib.check(lib.papc_sa_mlp_fwd(ctypes.byref(plan), ctypes.byref(io), st), "papc_sa_mlp_fwd")
    # the kernels' decisions, from the saved buffer
    def view(off, n, dt):
        return saved[off: off + 4 * n].view(dt)
    argmax = view(plan.off_argmax, G * chans[-1], torch.int32).view(G, chans[-1]).clone()
    masks = []
    for l in range(L - 1):
        if plan.off_y[l] < 0:
            masks.append(None)
            continue
        y = view(plan.off_y[l], Mrows * chans[l + 1], torch.float32).view(Mrows, chans[l + 1])
        cst = view(plan.off_cst[l], 4 * chans[l + 1], torch.float32).view(4, chans[l + 1])
        masks.append((cst[2].double() * y.double() + cst[3].double()) > 0)
    g = SaGrads()
    g.gout = gout.data_ptr()
    grads = []
    for l in range(L):
        cin = D + 3 if l == 0 else chans[l]
        t = [torch.empty(chans[l + 1], cin, device=dev), torch.empty(chans[l + 1], device=dev), torch.empty(chans[l + 1], device=dev),
             torch.empty(chans[l + 1], device=dev)]
        g.dw[l], g.db[l], g.dgamma[l], g.dbeta[l] = (x.data_ptr() for x in t)
        grads.append(t)
    gf = None
    if want_feats_grad:
        gf = torch.empty(B, N, D, device=dev)
        g.grad_feats = gf.data_ptr()
    _lib.check(lib.papc_sa_mlp_bwd(ctypes.byref(plan), ctypes.byref(io), ctypes.byref(g), st), "papc_sa_mlp_bwd")
    torch.cuda.synchronize()
    return out, grads, gf, plan, (argmax, masks)


@pytest.mark.parametrize("name,B,N,S,K,radius,D,chans", [
    ("SA1 (coordinates only: moment-path first layer, no-store max layer)", 8, 4096, 512, 32, 0.2, 0, [3, 64, 64, 128]),
    ("SA2 (gather-add first layer, compacted)", 8, 512, 128, 64, 0.4, 128, [131, 128, 128, 256]),
    ("SA2 (padded)", 8, 512, 128, 64, 0.4, 128, [131, 128, 128, 256]),
    ("small ragged stack (tiled kernels)", 3, 300, 20, 16, 0.3, 5, [8, 32, 48, 20]),
])
def test_stack_through_raw_c_entry_points_vs_f64(dev, name, B, N, S, K, radius, D, chans):
    xyz, new_xyz, idx = _sample(dev, B, N, S, K, radius, 21)
    rng = np.random.default_rng(4)
    feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev) if D else None
    ws = seeded_weights(chans, 77)
    gout = torch.from_numpy(rng.normal(size=(B * S, chans[-1])).astype(np.float32)).to(dev)
    cp = C.plan(idx) if "compacted" in name else None
    out, grads, gf, plan, (argmax, masks) = _raw_stack(dev, B, N, S, K, D, chans, xyz, new_xyz, feats, idx, ws, gout, cp, want_feats_grad=D > 0)
    print(name, "-> lin0 %d xyz1 %d gmax %d nostore %d compact %d; saved %.1f MB" % (plan.lin0, plan.xyz1, plan.gmax, plan.nostore, plan.compact,
                                                                                    plan.saved_bytes / 1e6))
    if "SA1" in name:
        assert plan.xyz1 and plan.nostore
    if "compacted" in name:
        assert plan.compact and plan.lin0
        start = cp.start.long()
        n = start[1:] - start[:-1]
        k = torch.arange(K, device=dev).view(1, -1)
        rowmap = (start[:-1].view(-1, 1) + torch.where(k < n.view(-1, 1), k, torch.zeros_like(k))).reshape(-1)
        argmax = (argmax.long() - start[:-1].view(-1, 1)).int()
        masks = [None if m is None else m[rowmap] for m in masks]
    # float64 reference on the same rows, routed through the kernels' own decisions
    p64 = [[torch.from_numpy(a).to(dev).double().requires_grad_(True) for a in tup] for tup in ws]
    f64 = feats.double().requires_grad_(True) if feats is not None else None
    rows = torch_ref.group(xyz.double(), new_xyz.double(), f64, idx, True).reshape(B * S * K, D + 3)
    ref, stats = torch_ref.stack_routed(rows, [tuple(t) for t in p64], K, 1e-5, argmax, out > 0, masks)
    print("decisions that differ from float64's own:", stats)
    assert_close(out.cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, name + ": forward")
    ref.backward(gout.double())
    for l in range(len(chans) - 1):
        for j, nm in ((0, "w"), (2, "gamma"), (3, "beta")):
            assert_close(grads[l][j].cpu().numpy(), p64[l][j].grad.cpu().numpy(), 2e-4, "%s: d%s layer %d" % (name, nm, l))
        assert float(grads[l][1].abs().max()) <= 1e-4 * float(p64[l][0].grad.abs().max())      # bias under a train-mode BN: ~0
    if gf is not None:
        assert_close(gf.cpu().numpy(), f64.grad.cpu().numpy(), 2e-4, name + ": dfeats")


def test_group_all_stack_through_raw_c_entry_points_vs_oracle(dev):
    """SA3 of the classifier (sample_and_group_all, :160-176) through the same three calls: identity rows, 259 -> 256 -> 512 -> 1024"""
    B, N, D = 4, 128, 256
    x = make_clouds(B, N, 5)
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(B, D, N)).astype(np.float32)
    chans = [D + 3, 256, 512, 1024]
    ws = seeded_weights(chans, 6)
    ora = R.PointNetSetAbstraction(None, None, None, D + 3, chans[1:], True, ws)
    _, ref64 = ora.forward(x, pts, f64=True)
    xyz = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
    feats = torch.from_numpy(np.ascontiguousarray(pts.transpose(0, 2, 1))).to(dev)
    gout = torch.zeros(B, 1024, device=dev)
    out, grads, gf, plan, _ = _raw_stack(dev, B, N, 1, N, D, chans, xyz, torch.zeros(B, 1, 3, device=dev), feats, None, ws, gout)
    assert plan.planes, "the group_all stack must take the planes kernels inside the library"
    assert_close(out.cpu().numpy().reshape(B, 1024), ref64.reshape(B, 1024), 1e-5, "group_all via the C entry points vs f64 oracle")


@pytest.mark.parametrize("shape", ["sa1", "sa1_fused", "sa2", "sa2_compact", "plain_nopool", "small"])
def test_library_orchestration_equals_the_python_launch_sequence(dev, shape, monkeypatch):
    """papc_sa_mlp_fwd / _bwd issue the launches papc_amd.mlp.SharedMLPMax spells out in Python: bit-identical results.  (sa1_fused: the
    library's own extra step -- the dX above the coordinates-only first layer folded into that layer's four sums instead of stored and
    read back, papc_mlp_bwd_dx_xyz_f32 -- changes the summation order of the first layer's three gradients only.)"""
    rng = np.random.default_rng(9)
    fused = shape == "sa1_fused"
    monkeypatch.setattr(M_, "_XYZ_FUSE", fused)
    # (the Python sequence keeps the compacted stack's statistics as unweighted sums + papc_bn_stats_corr_f32: hold the library to the same
    # form here; its default -- the producers weigh the sums themselves -- is held to this one in tests/test_gpu_compact.py)
    monkeypatch.setenv("PAPC_WSTATS", "0")
    if fused:
        shape = "sa1"
    feats = idx = x_rows = xyz = new_xyz = None
    if shape == "plain_nopool":
        Mr, chans = 70000, [40, 64, 32]
        x_rows = torch.from_numpy(rng.normal(size=(Mr, 40)).astype(np.float32)).to(dev)
        mk = lambda: StackSpec(1, Mr, Mr, 1, 37, True, pool=False)
        gshape = (Mr, 32)
    else:
        B, N, S, K, r, D, chans = {"sa1": (8, 4096, 512, 32, 0.2, 0, [3, 64, 64, 128]), "sa2": (8, 512, 128, 64, 0.4, 128, [131, 128, 128, 256]),
                                   "sa2_compact": (8, 512, 128, 64, 0.4, 128, [131, 128, 128, 256]), "small": (3, 300, 20, 16, 0.3, 5, [8, 32, 48, 20])}[shape]
        xyz, new_xyz, idx = _sample(dev, B, N, S, K, r, 33)
        if D:
            feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev)
        mk = lambda: StackSpec(B, N, S, K, D, True)
        gshape = (B * S, chans[-1])
    ws = seeded_weights(chans, 12)
    gout = torch.from_numpy(rng.normal(size=gshape).astype(np.float32)).to(dev)
    res = []
    for fn in (M_.SharedMLPMax, SharedMLPStack):
        spec = mk()
        if shape == "sa2_compact":
            spec.compact = C.plan(idx)
        params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        f = feats.clone().requires_grad_(True) if feats is not None else None
        xr = x_rows.clone().requires_grad_(True) if x_rows is not None else None
        out = fn.apply(spec, None, xyz, new_xyz, f, idx, xr, *params)
        out.backward(gout)
        res.append((out.detach(), [p.grad for p in params], None if f is None else f.grad, None if xr is None else xr.grad))
    (o0, g0, f0, x0), (o1, g1, f1, x1) = res
    assert torch.equal(o0, o1), "forward differs"
    for i, (a, b) in enumerate(zip(g0, g1)):
        assert (a is None) == (b is None)
        if a is None:
            continue
        if fused and i in (0, 2, 3):     # dW, dgamma, dbeta of the coordinates-only layer (its bias gradient is an exact zero)
            assert not torch.equal(a, b) or i != 0, "the fused dX did not run"
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), "gradient %d of the first layer differs beyond summation order" % i
        elif i == 0 and feats is not None and chans[1] % 4 == 0 and feats.shape[2] >= 16:
            # the gather-add first layer's dW_f = G^T feats reads G, the float-atomic row sums: same terms, run-dependent order
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), "gradient 0 differs beyond atomic-order noise"
        else:
            assert torch.equal(a, b), "gradient %d differs" % i
    if x0 is not None:
        assert torch.equal(x0, x1)
    if f0 is not None:      # (float atomics in the gather-add backward: same terms, run-dependent order)
        assert float((f0 - f1).abs().max()) <= 1e-5 * float(f0.abs().max())


def test_pfn_through_raw_c_entry_points_vs_oracle(dev):
    """PillarFeatureNet.forward (pillars.py:79-108) as papc_pfn_fwd / papc_pfn_bwd through raw ctypes: the pooled features against the float64
    oracle at 1e-5, the three parameter gradients against float64 torch autograd on the oracle's decorated rows at 2e-4"""
    from papc_amd.pillars import PfnDesc, PfnIo
    from papc_amd.synthetic import make_pillars
    lib = _lib.load()
    P, T, Cc = 1500, 100, 64
    voxels, nump, coors = make_pillars(P, T, seed=7)
    rng = np.random.default_rng(3)
    w = (rng.normal(size=(Cc, 9)) * 0.3).astype(np.float32)
    g = rng.uniform(0.5, 1.5, Cc).astype(np.float32) * rng.choice([1.0, -1.0], Cc).astype(np.float32)
    b = (rng.normal(size=Cc) * 0.2).astype(np.float32)
    vs, pr = (0.16, 0.16, 4.0), (0.0, -39.68, -3.0, 69.12, 39.68, 1.0)
    ref = R.pillar_feature_net(voxels, nump, coors, [(w, g, b)], vs, pr, f64=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    feat, nv, co, wt, gt, bt = t(voxels), t(nump), t(coors), t(w), t(g), t(b)
    d = PfnDesc(P, T, Cc, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1], 1e-3, 0.01, 1)
    sb, wb = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.papc_pfn_workspace(ctypes.byref(d), ctypes.byref(sb), ctypes.byref(wb)), "papc_pfn_workspace")
    saved = torch.empty(sb.value, device=dev, dtype=torch.uint8)
    scr = torch.empty(wb.value, device=dev, dtype=torch.uint8)
    out = torch.empty(P, Cc, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    io = PfnIo(feat.data_ptr(), nv.data_ptr(), co.data_ptr(), wt.data_ptr(), gt.data_ptr(), bt.data_ptr(), rm.data_ptr(), rv.data_ptr(), out.data_ptr(),
               saved.data_ptr(), scr.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.papc_pfn_fwd(ctypes.byref(d), ctypes.byref(io), st), "papc_pfn_fwd")
    assert_close(out.cpu().numpy(), np.asarray(ref).reshape(P, Cc), 1e-5, "papc_pfn_fwd vs f64 oracle")
    gout = torch.from_numpy(rng.normal(size=(P, Cc)).astype(np.float32)).to(dev)
    dw, dg, db = torch.empty(Cc, 9, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    _lib.check(lib.papc_pfn_bwd(ctypes.byref(d), ctypes.byref(io), gout.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, st), "papc_pfn_bwd")
    rows = torch.from_numpy(R.pillar_decorate(voxels, nump, coors, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1])).to(dev).double().reshape(P * T, 9)
    w64, g64, b64 = (x.double().requires_grad_(True) for x in (wt, gt, bt))
    y = rows @ w64.t()
    z = (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-3) * g64 + b64
    o64 = torch.relu(z).reshape(P, T, Cc).max(1).values
    o64.backward(gout.double())
    assert_close(dw.cpu().numpy(), w64.grad.cpu().numpy(), 2e-4, "papc_pfn_bwd dW")
    assert_close(dg.cpu().numpy(), g64.grad.cpu().numpy(), 2e-4, "papc_pfn_bwd dgamma")
    assert_close(db.cpu().numpy(), b64.grad.cpu().numpy(), 2e-4, "papc_pfn_bwd dbeta")
    assert abs(float(rm.abs().max())) > 0                                # running statistics were updated (paddle momentum 0.01 weighs the OLD value)


@pytest.mark.parametrize("pool", [True, False])
def test_library_planes_orchestration_equals_the_python_one(dev, pool):
    """few-row stacks (sample_and_group_all / point-wise): papc_sa_mlp_fwd / _bwd sequence the planes kernels (csrc/smallm.hip) exactly as
    papc_amd.smallm.PlanesMLPMax does in Python -- bit-identical outputs and gradients (no atomics on this path)"""
    from papc_amd import smallm
    rng = np.random.default_rng(2)
    if pool:
        B, N, D, chans = 8, 128, 256, [259, 256, 512, 1024]
        xyz = torch.from_numpy(np.ascontiguousarray(make_clouds(B, N, 5).transpose(0, 2, 1))).to(dev)
        feats0 = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev)
        args = lambda f: (StackSpec(B, N, 1, N, D, True), None, xyz, torch.zeros(B, 1, 3, device=dev), f, None, None)
        gshape = (B, 1024)
    else:
        Mr, chans = 2048, [576, 256, 128]
        x0 = torch.from_numpy(rng.normal(size=(Mr, 576)).astype(np.float32)).to(dev)
        args = lambda xr: (StackSpec(1, Mr, Mr, 1, 573, True, pool=False), None, None, None, None, None, xr)
        gshape = (Mr, 128)
    ws = seeded_weights(chans, 12)
    gout = torch.from_numpy(rng.normal(size=gshape).astype(np.float32)).to(dev)
    res = []
    for fn in (smallm.PlanesMLPMax, SharedMLPStack):
        params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        inp = (feats0 if pool else x0).clone().requires_grad_(True)
        out = fn.apply(*args(inp), *params)
        if fn is SharedMLPStack:
            assert out.grad_fn.planes
        out.backward(gout)
        res.append((out.detach(), [p.grad for p in params], inp.grad))
    (o0, g0, i0), (o1, g1, i1) = res
    assert torch.equal(o0, o1) and torch.equal(i0, i1)
    for a, b in zip(g0, g1):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))


def test_pfn_fused_tails_equal_the_separate_launches(dev):
    """papc_pfn_fwd / _bwd with the BatchNorm constants and the dW finalize as the last-arriving workgroup's tail of the Gram pass / the fold
    (PAPC_PFN_FUSED_TAILS=1, the default: 4 launches per frame) against the separate finalize launches (=0: 6): the same sums in the same
    order, so every output is bit-identical -- over several frames in a row (the ticket words must return to zero) and at the full KITTI
    frame size of BASELINE configs[4]."""
    from papc_amd.pillars import PfnDesc, PfnIo
    from papc_amd.synthetic import make_pillars
    lib = _lib.load()

    def run(P, T, Cc, seed, fused, reps, zero_padded=0):
        if True:
            voxels, nump, coors = make_pillars(P, T, seed=seed)
            rng = np.random.default_rng(seed)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            feat, nv, co = t(voxels), t(nump), t(coors)
            wt = t((rng.normal(size=(Cc, 9)) * 0.3).astype(np.float32))
            gt = t(rng.uniform(0.5, 1.5, Cc).astype(np.float32))
            bt = t((rng.normal(size=Cc) * 0.2).astype(np.float32))
            gout = t(rng.normal(size=(P, Cc)).astype(np.float32))
            d = PfnDesc(P, T, Cc, 0.16, 0.16, 0.08, -39.6, 1e-3, 0.01, 1, zero_padded)
            sb, wb = ctypes.c_int64(0), ctypes.c_int64(0)
            _lib.check(lib.papc_pfn_workspace(ctypes.byref(d), ctypes.byref(sb), ctypes.byref(wb)), "papc_pfn_workspace")
            saved = torch.empty(sb.value, device=dev, dtype=torch.uint8)
            scr = torch.empty(wb.value, device=dev, dtype=torch.uint8)
            out = torch.empty(P, Cc, device=dev)
            rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
            tickets = torch.zeros(2, device=dev, dtype=torch.int32)      # caller-owned ticket words (papc_pfn_io.tickets): NULL = the separate finalize launches
            io = PfnIo(feat.data_ptr(), nv.data_ptr(), co.data_ptr(), wt.data_ptr(), gt.data_ptr(), bt.data_ptr(), rm.data_ptr(), rv.data_ptr(), out.data_ptr(),
                       saved.data_ptr(), scr.data_ptr(), tickets.data_ptr() if fused else None)
            st = torch.cuda.current_stream().cuda_stream
            res = []
            for r in range(reps):
                dw, dg, db = torch.empty(Cc, 9, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
                _lib.check(lib.papc_pfn_fwd(ctypes.byref(d), ctypes.byref(io), st), "papc_pfn_fwd")
                _lib.check(lib.papc_pfn_bwd(ctypes.byref(d), ctypes.byref(io), gout.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, st), "papc_pfn_bwd")
                torch.cuda.synchronize()
                res.append([x.clone() for x in (out, dw, dg, db, rm, rv)])
                wt.mul_(1.01)          # (another frame's weights: the statistics must be recomputed, not remembered)
            assert int(tickets.abs().sum()) == 0, "the ticket words did not return to zero"
            return res

    for P, T, Cc, reps in ((1500, 100, 64, 4), (12000, 100, 64, 2), (37, 20, 16, 3)):
        a = run(P, T, Cc, 11, 1, reps)
        b = run(P, T, Cc, 11, 0, reps)
        for ra, rb in zip(a, b):
            for xa, xb, nm in zip(ra, rb, ("out", "dw", "dgamma", "dbeta", "running_mean", "running_var")):
                assert torch.isfinite(xa).all(), nm
                assert torch.equal(xa, xb), "fused tails: %s differs (P=%d)" % (nm, P)
        # zero_padded (papc_pfn_desc): rows behind num_voxels are zero in these frames -- the reference's voxeliser zero-initialises its
        # buffers (libs/ops/point_cloud/point_cloud_ops.py:148) -- so loading only the real rows changes nothing, bit for bit
        c = run(P, T, Cc, 11, 1, reps, zero_padded=1)
        for ra, rc in zip(a, c):
            for xa, xc, nm in zip(ra, rc, ("out", "dw", "dgamma", "dbeta", "running_mean", "running_var")):
                assert torch.equal(xa, xc), "zero_padded: %s differs (P=%d)" % (nm, P)
