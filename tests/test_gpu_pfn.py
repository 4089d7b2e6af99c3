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
