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
