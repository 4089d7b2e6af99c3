"""GPU parity of PillarFeatureNet / PFNLayer against the oracle (pillars.py:9-108) and a torch autograd reference."""
import os

import numpy as np
import pytest
import torch

from oracle import reference_np as R
from papc_amd.pillars import PFNLayer, PillarFeatureNet
from papc_amd.synthetic import make_pillars
from tests.util import assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _weights(C, cin, seed):
    rng = np.random.default_rng(seed)
    w = (rng.normal(size=(C, cin)) * 0.3).astype(np.float32)
    g = rng.uniform(0.5, 1.5, C).astype(np.float32) * rng.choice([1.0, 1.0, -1.0], C).astype(np.float32)
    b = (rng.normal(size=C) * 0.1).astype(np.float32)
    return w, g, b


def _load(pfn, w, g, b):
    with torch.no_grad():
        pfn.linear.weight.copy_(torch.from_numpy(w)); pfn.norm.weight.copy_(torch.from_numpy(g)); pfn.norm.bias.copy_(torch.from_numpy(b))


@pytest.mark.parametrize("vs,pr", [((1, 2, 3), (0, -40, -3, 70.4, 40, 1)),                    # the reference's own wiring
                                   ((0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1))])      # the voxeliser's grid
@pytest.mark.parametrize("P,T", [(64, 100), (257, 33), (5, 128)])
def test_pfn_forward_vs_oracle(dev, vs, pr, P, T):
    voxels, nump, coors = make_pillars(P=P, T=T, seed=11)
    nump[0] = 1
    nump[1] = T
    voxels *= (np.arange(T)[None, :] < nump[:, None])[:, :, None]
    w, g, b = _weights(64, 9, 3)
    ref = R.pillar_feature_net(voxels, nump, coors, [(w, g, b)], vs, pr, f64=True)
    net = PillarFeatureNet(num_filters=(64,), voxel_size=vs, pc_range=pr).to(dev)
    _load(net.pfn_layers[0], w, g, b)
    got = net(torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev))
    assert tuple(got.shape) == (P, 64)
    assert_close(got.detach().cpu().numpy(), ref, 1e-5, "PFN vs f64 oracle")
    dec = net.decorate(torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev))
    refd = R.pillar_decorate(voxels, nump, coors, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1])
    assert np.array_equal(dec.cpu().numpy()[..., :4], refd[..., :4])          # raw channels and mask: exact
    assert np.array_equal(dec.cpu().numpy()[..., 7:], refd[..., 7:])          # pillar-centre offsets: exact
    assert np.allclose(dec.cpu().numpy()[..., 4:7], refd[..., 4:7], rtol=0, atol=2e-5)   # cluster mean: summation order


@pytest.mark.parametrize("F,T,filters,with_distance", [(5, 100, (64,), False), (3, 40, (32, 64), False), (6, 150, (64,), False), (4, 150, (64,), False),
                                                       (5, 20, (16,), True)])
def test_pfn_other_point_widths(dev, F, T, filters, with_distance):
    # num_input_features != 4 (e.g. [x, y, z, r, t] sweeps) and T > 128: the general decoration + the PFNLayer stack, vs the oracle
    P = 97
    voxels4, nump, coors = make_pillars(P=P, T=T, seed=5 + F)
    rng = np.random.default_rng(F)
    extra = rng.normal(size=(P, T, max(F - 3, 0))).astype(np.float32)
    voxels = np.concatenate([voxels4[..., :3], extra], -1)
    voxels *= (np.arange(T)[None, :] < nump[:, None])[:, :, None]
    vs, pr = (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1)
    net = PillarFeatureNet(num_input_features=F, num_filters=filters, with_distance=with_distance, voxel_size=vs, pc_range=pr).to(dev)
    ws, cin = [], F + 5 + (1 if with_distance else 0)
    for i, pfn in enumerate(net.pfn_layers):
        w, g, b = _weights(pfn.units, cin, 7 + i)
        _load(pfn, w, g, b)
        ws.append((w, g, b))
        cin = 2 * pfn.units
    tv, tn, tc = torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev)
    dec = net.decorate(tv, tn, tc).cpu().numpy()
    refd = R.pillar_decorate(voxels, nump, coors, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1], with_distance=with_distance)
    assert dec.shape == refd.shape
    assert np.array_equal(dec[..., :F], refd[..., :F]) and np.array_equal(dec[..., F + 3:F + 5], refd[..., F + 3:F + 5])
    # cluster mean: two fp32 summation orders (numpy's pairwise blocks, the kernel's lane-then-wave tree), each within
    # ~log2(T) eps max|x| of the exact mean
    tol = 2 * np.finfo(np.float32).eps * np.log2(T) * np.abs(voxels[..., :3]).max()
    assert np.abs(dec[..., F:F + 3] - refd[..., F:F + 3]).max() <= max(tol, 2e-5)
    if with_distance:
        assert np.allclose(dec[..., F + 5], refd[..., F + 5], rtol=1e-6, atol=0)
        return                                                                                 # (the oracle's net has no with_distance switch)
    got = net(tv, tn, tc)
    ref = R.pillar_feature_net(voxels, nump, coors, ws, vs, pr, f64=True)
    assert_close(got.detach().cpu().numpy(), ref, 1e-5, "PFN F=%d vs f64 oracle" % F)
    got.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_pfn_golden(dev):
    g = np.load(os.path.join(GOLD, "pfn_p64.npz"))
    for tag, vs, pr in [("ref_wiring", (1, 2, 3), (0, -40, -3, 70.4, 40, 1)), ("voxel_wiring", (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1))]:
        net = PillarFeatureNet(num_filters=(64,), voxel_size=vs, pc_range=pr).to(dev)
        _load(net.pfn_layers[0], g["w"], g["gamma"], g["beta"])
        got = net(torch.from_numpy(g["voxels"]).to(dev), torch.from_numpy(g["num_points"]).to(dev), torch.from_numpy(g["coors"]).to(dev))
        assert_close(got.detach().cpu().numpy(), g["out_" + tag], 1e-5, "PFN golden " + tag)


def test_pfn_backward_vs_torch(dev):
    P, T = 200, 40
    voxels, nump, coors = make_pillars(P=P, T=T, seed=5)
    w, g, b = _weights(64, 9, 7)
    vs, pr = (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1)
    net = PillarFeatureNet(num_filters=(64,), voxel_size=vs, pc_range=pr).to(dev)
    _load(net.pfn_layers[0], w, g, b)
    out = net(torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev))
    gout = torch.randn_like(out)
    out.backward(gout)
    # float64 torch reference on the oracle's decorated rows
    rows = torch.from_numpy(R.pillar_decorate(voxels, nump, coors, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1])).to(dev).double()
    w64 = torch.from_numpy(w).to(dev).double().requires_grad_(True)
    g64 = torch.from_numpy(g).to(dev).double().requires_grad_(True)
    b64 = torch.from_numpy(b).to(dev).double().requires_grad_(True)
    y = rows.reshape(P * T, 9) @ w64.t()
    z = torch.relu((y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-3) * g64 + b64).reshape(P, T, 64).max(1).values
    assert_close(out.detach().cpu().numpy(), z.detach().cpu().numpy(), 1e-5, "PFN fwd")
    z.backward(gout.double())
    lay = net.pfn_layers[0]
    assert_close(lay.linear.weight.grad.cpu().numpy(), w64.grad.cpu().numpy(), 2e-4, "PFN dW")
    assert_close(lay.norm.weight.grad.cpu().numpy(), g64.grad.cpu().numpy(), 2e-4, "PFN dgamma")
    assert_close(lay.norm.bias.grad.cpu().numpy(), b64.grad.cpu().numpy(), 2e-4, "PFN dbeta")


def test_pfn_two_layer_chain_and_standalone_layer(dev):
    P, T = 96, 50
    voxels, nump, coors = make_pillars(P=P, T=T, seed=2)
    w1, g1, b1 = _weights(32, 9, 1)
    w2, g2, b2 = _weights(128, 64, 2)
    ref = R.pillar_feature_net(voxels, nump, coors, [(w1, g1, b1), (w2, g2, b2)], f64=True)
    net = PillarFeatureNet().to(dev)                       # default num_filters=(64,128)
    _load(net.pfn_layers[0], w1, g1, b1)
    _load(net.pfn_layers[1], w2, g2, b2)
    got = net(torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev))
    assert tuple(got.shape) == (P, 128)
    assert_close(got.detach().cpu().numpy(), ref, 1e-5, "two-layer PFN vs f64 oracle")
    lay = PFNLayer(9, 64, last_layer=True).to(dev)
    _load(lay, *_weights(64, 9, 4))
    x = torch.randn(P, T, 9, device=dev)
    o = lay(x)
    assert tuple(o.shape) == (P, 1, 64)
    refl = R.pfn_layer(x.cpu().numpy(), *_weights(64, 9, 4), last_layer=True, f64=True)
    assert_close(o.detach().cpu().numpy(), refl, 1e-5, "PFNLayer vs oracle")


def test_pfn_full_size_config5(dev):
    """BASELINE config 5: 12000 pillars x 100 points.  Properties: permuting pillars permutes rows; output >= 0;
    empty-padded points never win unless relu(shift-like value) does (checked through the oracle on a slice)."""
    voxels, nump, coors = make_pillars()
    w, g, b = _weights(64, 9, 3)
    net = PillarFeatureNet(num_filters=(64,), voxel_size=(0.16, 0.16, 4), pc_range=(0, -39.68, -3, 69.12, 39.68, 1)).to(dev)
    _load(net.pfn_layers[0], w, g, b)
    tv, tn, tc = torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev)
    out = net(tv, tn, tc)
    assert tuple(out.shape) == (12000, 64) and (out >= 0).all() and torch.isfinite(out).all()
    perm = torch.randperm(12000, device=dev)
    out_p = net(tv[perm].contiguous(), tn[perm].contiguous(), tc[perm].contiguous())
    assert_close(out_p.cpu().detach().numpy(), out[perm].cpu().detach().numpy(), 1e-5, "pillar permutation equivariance")
    ref = R.pillar_feature_net(voxels, nump, coors, [(w, g, b)], (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1), f64=True)
    assert_close(out.detach().cpu().numpy(), ref, 1e-5, "config-5 PFN vs f64 oracle")


def _torch_pfn_layer(x, w, g, b, bias, last, use_norm, eps=1e-3, running=None):
    """float64 torch restatement of PFNLayer.forward (pillars.py:29-41) on [P,T,Cin] rows"""
    P, T, _ = x.shape
    y = x.reshape(P * T, -1) @ w.t()
    if bias is not None:
        y = y + bias
    if use_norm:
        if running is None:
            mean, var = y.mean(0), y.var(0, unbiased=False)
        else:
            mean, var = running
        y = (y - mean) / torch.sqrt(var + eps) * g + b
    z = torch.relu(y).reshape(P, T, -1)
    zmax = z.max(1, keepdim=True).values
    return zmax if last else torch.cat([z, zmax.expand(P, T, zmax.shape[2])], 2)


@pytest.mark.parametrize("use_norm,with_distance", [(True, False), (False, False), (True, True), (False, True)])
def test_pfn_default_ctor_trains_every_layer(dev, use_norm, with_distance):
    """The source's DEFAULT PillarFeatureNet(num_filters=(64, 128)) (pillars.py:46): forward and the gradients of BOTH layers
    (the non-last layer returns its activations concatenated with the tiled max, :39-41) against a float64 torch reference;
    likewise use_norm=False (:25-27) and with_distance=True (:57-58, :92-94)."""
    P, T = 150, 40
    voxels, nump, coors = make_pillars(P=P, T=T, seed=8)
    vs, pr = (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1)
    net = PillarFeatureNet(use_norm=use_norm, with_distance=with_distance, voxel_size=vs, pc_range=pr).to(dev)
    cin = 10 if with_distance else 9
    rng = np.random.default_rng(3)
    ws = [_weights(32, cin, 1), _weights(128, 64, 2)]
    bs = [(rng.normal(size=32) * 0.1).astype(np.float32), (rng.normal(size=128) * 0.1).astype(np.float32)]
    with torch.no_grad():
        for lay, (w, g, b), bias in zip(net.pfn_layers, ws, bs):
            lay.linear.weight.copy_(torch.from_numpy(w))
            if use_norm:
                lay.norm.weight.copy_(torch.from_numpy(g)); lay.norm.bias.copy_(torch.from_numpy(b))
            else:
                lay.linear.bias.copy_(torch.from_numpy(bias))
    out = net(torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev))
    assert tuple(out.shape) == (P, 128)
    gout = torch.randn_like(out)
    out.backward(gout)
    rows = torch.from_numpy(R.pillar_decorate(voxels, nump, coors, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1],
                                              with_distance=with_distance)).to(dev).double()
    dec = net.decorate(torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev))
    assert np.allclose(dec.cpu().numpy(), rows.cpu().numpy(), rtol=0, atol=2e-5)
    if with_distance:
        assert_close(dec.cpu().numpy()[..., 9], rows.cpu().numpy()[..., 9], 2e-7, "points_dist channel")
    p64 = []
    x = rows
    for i, ((w, g, b), bias) in enumerate(zip(ws, bs)):
        t = [torch.from_numpy(a).to(dev).double().requires_grad_(True) for a in (w, g, b, bias)]
        p64.append(t)
        x = _torch_pfn_layer(x, t[0], t[1], t[2], None if use_norm else t[3], i == 1, use_norm)
    ref = x.squeeze()
    assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "default-ctor PFN forward")
    ref.backward(gout.double())
    for i, lay in enumerate(net.pfn_layers):
        assert_close(lay.linear.weight.grad.cpu().numpy(), p64[i][0].grad.cpu().numpy(), 2e-4, "dW layer %d" % i)
        if use_norm:
            assert_close(lay.norm.weight.grad.cpu().numpy(), p64[i][1].grad.cpu().numpy(), 2e-4, "dgamma layer %d" % i)
            assert_close(lay.norm.bias.grad.cpu().numpy(), p64[i][2].grad.cpu().numpy(), 2e-4, "dbeta layer %d" % i)
        else:
            assert_close(lay.linear.bias.grad.cpu().numpy(), p64[i][3].grad.cpu().numpy(), 2e-4, "db layer %d" % i)


@pytest.mark.parametrize("filters", [(64,), (64, 128)])
def test_pfn_eval_uses_running_statistics(dev, filters):
    """model.eval(): the norms are registered layers in the source (pillars.py:24), so inference normalises with the running
    statistics, does not depend on the batch composition and leaves the statistics untouched."""
    P, T = 120, 30
    voxels, nump, coors = make_pillars(P=P, T=T, seed=4)
    vs, pr = (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1)
    net = PillarFeatureNet(num_filters=filters, voxel_size=vs, pc_range=pr).to(dev)
    tv, tn, tc = torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev)
    net.train()
    for _ in range(3):
        net(tv, tn, tc)                                    # move the running statistics away from (0, 1)
    net.eval()
    before = [(l.norm.running_mean.clone(), l.norm.running_var.clone()) for l in net.pfn_layers]
    out = net(tv, tn, tc)
    half = net(tv[: P // 2].contiguous(), tn[: P // 2].contiguous(), tc[: P // 2].contiguous())
    assert_close(half.detach().cpu().numpy(), out[: P // 2].detach().cpu().numpy(), 1e-6, "eval output independent of the batch")
    for l, (m, v) in zip(net.pfn_layers, before):
        assert torch.equal(l.norm.running_mean, m) and torch.equal(l.norm.running_var, v)
    x = torch.from_numpy(R.pillar_decorate(voxels, nump, coors, vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1])).to(dev).double()
    for i, l in enumerate(net.pfn_layers):
        x = _torch_pfn_layer(x, l.linear.weight.double(), l.norm.weight.double(), l.norm.bias.double(), None, i == len(filters) - 1, True,
                             running=(l.norm.running_mean.double(), l.norm.running_var.double()))
    assert_close(out.detach().cpu().numpy(), x.squeeze().detach().cpu().numpy(), 1e-5, "eval PFN vs f64 torch")


def test_pfn_gram_path_matches_two_pass_kernels(dev):
    """The Gram path (BN statistics and dW from the inputs' 10x10 Gram matrix, float64 MFMA) against the dense two-pass entry points
    (papc_pfn_stats_f32 + papc_bn_finalize_f32, papc_pfn_bwd_reduce_f32 + papc_bn_bwd_finalize_f32 + papc_pfn_bwd_dw_f32) on the full
    config-5 frame (coordinates up to 70 m: the quadratic forms cancel), and the Gram matrix itself against float64 numpy."""
    import ctypes
    from papc_amd import _lib
    lib = _lib.load()
    P, T, C = 12000, 100, 64
    voxels, nump, coors = make_pillars()
    w, g, b = _weights(C, 9, 3)
    vs, pr = (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1)
    geo = (vs[0], vs[1], vs[0] / 2 + pr[0], vs[1] / 2 + pr[1])
    tv, tn, tc = torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev)
    tw, tg, tb = torch.from_numpy(w).to(dev), torch.from_numpy(g).to(dev), torch.from_numpy(b).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    # dense two-pass statistics
    nb = lib.papc_pfn_num_blocks(P)
    stats = torch.empty(nb, 2, C, device=dev)
    cst_a = torch.empty(4, C, device=dev)
    _lib.check(lib.papc_pfn_stats_f32(p(tv), p(tn), p(tc), P, T, *geo, p(tw), C, p(stats), None, None), "stats")
    _lib.check(lib.papc_bn_finalize_f32(p(stats), nb, P * T, C, p(tg), p(tb), 1e-3, 0.9, p(cst_a[0]), p(cst_a[1]), p(cst_a[2]), p(cst_a[3]),
                                        None, None, None), "finalize")
    # Gram path
    ng = lib.papc_pfn_gram_blocks(P)
    gpart = torch.empty(ng, 256, device=dev, dtype=torch.float64)
    gram = torch.empty(256, device=dev, dtype=torch.float64)
    cst_b = torch.empty(4, C, device=dev)
    _lib.check(lib.papc_pfn_gram_f32(p(tv), p(tn), p(tc), P, T, *geo, p(gpart), None), "gram")
    _lib.check(lib.papc_pfn_gram_finalize_f32(p(gpart), ng, P * T, p(tw), C, p(tg), p(tb), 1e-3, 0.9, p(cst_b[0]), p(cst_b[1]), p(cst_b[2]),
                                              p(cst_b[3]), None, None, p(gram), None), "gram finalize")
    rows = R.pillar_decorate(voxels, nump, coors, *geo).reshape(P * T, 9).astype(np.float64)
    X = np.concatenate([rows, np.ones((P * T, 1))], axis=1)
    G = gram.cpu().numpy().reshape(16, 16)
    idx = list(range(9)) + [10]
    Gref = X.T @ X
    assert np.abs(G[np.ix_(idx, idx)] - Gref).max() <= 1e-5 * np.abs(Gref).max()   # (the cluster means differ in summation order: 2e-5 absolute on a row)
    for i, name in enumerate(["mean", "invstd", "scale", "shift"]):
        assert_close(cst_b[i].cpu().numpy(), cst_a[i].cpu().numpy(), 2e-5, "Gram vs dense " + name)
    y64 = rows @ w.astype(np.float64).T
    assert_close(cst_b[0].cpu().numpy(), y64.mean(0), 1e-5, "Gram mean vs f64")
    assert_close(cst_b[1].cpu().numpy(), 1.0 / np.sqrt(y64.var(0) + 1e-3), 1e-5, "Gram invstd vs f64")
    # backward: same gout / argmax through both
    out = torch.empty(P, C, device=dev)
    am = torch.empty(P, C, device=dev, dtype=torch.int32)
    _lib.check(lib.papc_pfn_apply_f32(p(tv), p(tn), p(tc), P, T, *geo, p(tw), C, p(cst_a[2]), p(cst_a[3]), p(out), p(am), None), "apply")
    torch.manual_seed(0)
    gout = torch.randn(P, C, device=dev)
    bn = (p(cst_a[0]), p(cst_a[1]), p(cst_a[2]), p(cst_a[3]))
    red = torch.empty(nb, 2, C, device=dev)
    dgb_a, c12 = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    _lib.check(lib.papc_pfn_bwd_reduce_f32(p(tv), p(tn), p(tc), P, T, *geo, p(tw), C, p(gout), p(am), *bn, p(red), None), "bwd reduce")
    _lib.check(lib.papc_bn_bwd_finalize_f32(p(red), nb, P * T, C, p(dgb_a[0]), p(dgb_a[1]), p(c12[0]), p(c12[1]), 0, None), "bwd finalize")
    dwp = torch.empty(nb, C, 9, device=dev)
    dw_a = torch.empty(C, 9, device=dev)
    _lib.check(lib.papc_pfn_bwd_dw_f32(p(tv), p(tn), p(tc), P, T, *geo, p(tw), C, p(gout), p(am), *bn, p(c12[0]), p(c12[1]), p(dwp), None), "dw")
    _lib.check(lib.papc_reduce_partials_f32(p(dwp), nb, C * 9, p(dw_a), 0, None), "reduce")
    part = torch.empty(nb, 11, C, device=dev)
    sums = torch.empty(11, C, device=dev)
    dgb_b, dw_b = torch.empty(2, C, device=dev), torch.empty(C, 9, device=dev)
    _lib.check(lib.papc_pfn_bwd_sparse_f32(p(tv), p(tn), p(tc), P, T, *geo, p(tw), C, p(gout), p(am), *bn, p(part), None), "sparse")
    _lib.check(lib.papc_reduce_partials_f32(p(part), nb, 11 * C, p(sums), 0, None), "reduce")
    _lib.check(lib.papc_pfn_bwd_finalize_f32(p(sums), P * T, p(tw), C, p(gram), p(cst_a[0]), p(cst_a[1]), p(cst_a[2]), p(dgb_b[0]), p(dgb_b[1]),
                                             p(dw_b), 0, None), "gram bwd finalize")
    assert_close(dgb_b.cpu().numpy(), dgb_a.cpu().numpy(), 1e-5, "dgamma / dbeta")
    assert_close(dw_b.cpu().numpy(), dw_a.cpu().numpy(), 2e-4, "Gram dW vs dense dW")


@pytest.mark.parametrize("P,T,C", [(12000, 100, 64), (37, 128, 64), (64, 33, 48), (5, 1, 64)])
def test_pfn_apply_mfma_matches_valu_flavour(dev, P, T, C):
    """papc_pfn_apply_f32 on the bf16 matrix pipe (exact 3-way split, PAPC_PFN_MFMA=1, the default) against the lanes-are-channels
    fp32 fma-chain flavour: same max values to fp32 rounding, same argmax rows except where two rows tie to rounding, ragged T, C < 64,
    negative BN scales."""
    import ctypes
    from papc_amd import _lib
    lib = _lib.load()
    voxels, nump, coors = make_pillars(P=P, T=T, seed=21)
    w, g, b = _weights(C, 9, 5)
    geo = (0.16, 0.16, 0.08, -39.6)
    tv, tn, tc = torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev)
    tw = torch.from_numpy(w).to(dev)
    sc, sh = torch.from_numpy(g).to(dev) * 0.7, torch.from_numpy(b).to(dev) - 0.2
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    res = {}
    try:
        for flav in (0, 1):
            _lib.check(lib.papc_knob_set(b"PAPC_PFN_MFMA", flav), "knob")
            out = torch.full((P, C), -7.0, device=dev)
            am = torch.full((P, C), -7, device=dev, dtype=torch.int32)
            _lib.check(lib.papc_pfn_apply_f32(p(tv), p(tn), p(tc), P, T, *geo, p(tw), C, p(sc), p(sh), p(out), p(am), None), "apply")
            res[flav] = (out.cpu().numpy(), am.cpu().numpy())
    finally:
        _lib.check(lib.papc_knob_set(b"PAPC_PFN_MFMA", 1), "knob")
    (o0, a0), (o1, a1) = res[0], res[1]
    assert (o1 >= 0).all() and (a1 >= 0).all() and (a1 < T).all()
    assert_close(o1, o0, 2e-6, "MFMA apply vs VALU apply")
    differ = a0 != a1
    assert differ.mean() <= 2e-3, differ.mean()          # argmax rows: equal except at rounding-level ties
    if differ.any():                                     # ... and where they differ the two rows' values tie to rounding
        rows = R.pillar_decorate(voxels, nump, coors, *geo).astype(np.float64)
        pi, ci = np.nonzero(differ)
        y0 = np.einsum("nk,nk->n", rows[pi, a0[pi, ci]], w[ci].astype(np.float64))
        y1 = np.einsum("nk,nk->n", rows[pi, a1[pi, ci]], w[ci].astype(np.float64))
        z0, z1 = y0 * sc.cpu().numpy()[ci] + sh.cpu().numpy()[ci], y1 * sc.cpu().numpy()[ci] + sh.cpu().numpy()[ci]
        assert np.all(np.abs(z0 - z1) <= 1e-5 * (1 + np.abs(z0)) + (np.maximum(z0, z1) <= 1e-6))   # (both dead -> row 0 either way)


def test_pfn_inplace_gradients_match_autograd(dev):
    """FlatParams opts the parameters in to in-place accumulation: the fused PFN backward then ADDS dW / dgamma / dbeta straight into the
    flat gradient buffer (no AccumulateGrad kernels).  Same values as the autograd path, and a second backward accumulates."""
    from papc_amd.distributed import FlatParams
    P, T = 300, 50
    voxels, nump, coors = make_pillars(P=P, T=T, seed=9)
    w, g, b = _weights(64, 9, 2)
    vs, pr = (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1)
    tv, tn, tc = torch.from_numpy(voxels).to(dev), torch.from_numpy(nump).to(dev), torch.from_numpy(coors).to(dev)
    gout = torch.randn(P, 64, device=dev)
    nets = []
    for flat_mode in (False, True):
        net = PillarFeatureNet(num_filters=(64,), voxel_size=vs, pc_range=pr).to(dev)
        _load(net.pfn_layers[0], w, g, b)
        flat = FlatParams(net) if flat_mode else None
        for _ in range(2):
            net(tv, tn, tc).backward(gout)
        nets.append((net, flat))
    (ref, _), (got, flat) = nets
    for pr_, pg in zip(ref.parameters(), got.parameters()):
        assert pg.grad.data_ptr() >= flat.grad.data_ptr() and pg.grad.data_ptr() < flat.grad.data_ptr() + 4 * flat.grad.numel()
        assert_close(pg.grad.cpu().numpy(), pr_.grad.cpu().numpy(), 1e-6, "in-place PFN gradient")
