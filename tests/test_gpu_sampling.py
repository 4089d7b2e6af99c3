"""GPU parity of the index-exact kernels against the oracle, through the C ABI (via papc_amd.functional)."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R
from papc_amd import functional as F
from papc_amd.synthetic import make_clouds, make_start_idx

pytestmark = pytest.mark.gpu


def _cloud(B, N, seed):
    x = make_clouds(B, N, seed)                      # [B,3,N]
    return x, np.ascontiguousarray(x.transpose(0, 2, 1))


@pytest.mark.parametrize("B,N,npoint", [(2, 1024, 128), (1, 4096, 512), (3, 1000, 77), (2, 64, 64), (2, 37, 5), (1, 8192, 64),
                                        (1, 16384, 32), (4, 512, 128), (1, 1, 1), (2, 513, 40), (2, 1023, 64), (2, 1025, 64), (3, 300, 60),
                                        (2, 100, 100), (16, 2048, 512), (1, 5000, 33),
                                        # beyond the register-resident kernel (fps_big_kernel: 32 / 64 / 128 points per thread)
                                        (2, 16385, 16), (1, 30000, 40), (2, 50000, 24), (1, 131072, 12)])
@pytest.mark.parametrize("planar", [False, True])
def test_fps_index_exact(dev, B, N, npoint, planar):
    x, xyz = _cloud(B, N, 11 + N)
    st = make_start_idx(B, N, 5)
    ref = R.farthest_point_sample(xyz, npoint, st)
    t = torch.from_numpy(x).to(dev).transpose(1, 2) if planar else torch.from_numpy(xyz).to(dev)
    got = F.farthest_point_sample(t, npoint, torch.from_numpy(st).to(dev))
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), ref.astype(np.int64))


def test_fps_ties_and_init(dev):
    # engineered ties: many points at distance >= 1 from the start keep the initial 1.0 -> lowest index must win
    rng = np.random.default_rng(0)
    N = 2048
    v = rng.normal(size=(N, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    xyz = v.astype(np.float32)[None]                  # sphere shell radius 1: most pairs are > 1 apart
    xyz[0, 5] = xyz[0, 900]                           # duplicates
    st = np.array([17], np.int64)
    for init in (1.0, 1e10):
        ref = R.farthest_point_sample(xyz, 256, st, init)
        got = F.farthest_point_sample(torch.from_numpy(xyz).to(dev), 256, torch.from_numpy(st).to(dev), init_dist=init)
        assert np.array_equal(got.cpu().numpy(), ref.astype(np.int64)), init
    f = F.farthest_point_sample(torch.from_numpy(xyz).to(dev), 8, torch.from_numpy(st).to(dev), as_float=True)
    assert f.dtype == torch.float32                   # the source returns float32 centroids (:74)


@pytest.mark.parametrize("side,B", [(16, 2), (8, 3), (10, 1), (13, 2)])
def test_fps_lattice_ties(dev, side, B):
    # a shuffled cubic lattice: the running distances tie exactly among MANY distinct points in every iteration (equal squared
    # distances to the chosen set), across lanes, DPP rows and waves -- the first maximum in index order must win every time
    rng = np.random.default_rng(side)
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.125)
    xyz = np.stack([g[rng.permutation(len(g))] for _ in range(B)])
    st = rng.integers(0, len(g), size=B).astype(np.int64)
    npoint = min(len(g), 200)
    for init in (1e10, 1.0):
        ref = R.farthest_point_sample(xyz, npoint, st, init)
        got = F.farthest_point_sample(torch.from_numpy(xyz).to(dev), npoint, torch.from_numpy(st).to(dev), init_dist=init)
        assert np.array_equal(got.cpu().numpy(), ref.astype(np.int64)), (side, init)


def test_fps_all_duplicates(dev):
    xyz = np.ones((2, 300, 3), np.float32) * 0.25
    st = np.array([7, 299], np.int64)
    ref = R.farthest_point_sample(xyz, 10, st)
    got = F.farthest_point_sample(torch.from_numpy(xyz).to(dev), 10, torch.from_numpy(st).to(dev))
    assert np.array_equal(got.cpu().numpy(), ref.astype(np.int64))


@pytest.mark.parametrize("B,N,S", [(2, 1024, 128), (1, 4096, 512), (2, 777, 50), (1, 9000, 70)])
@pytest.mark.parametrize("radius,nsample", [(0.1, 16), (0.2, 32), (0.4, 64), (0.8, 128)])
def test_ball_query_exact(dev, B, N, S, radius, nsample):
    x, xyz = _cloud(B, N, 3 + N)
    st = make_start_idx(B, N, 9)
    new_xyz = R.index_points(xyz, R.farthest_point_sample(xyz, S, st))
    ref = R.query_ball_point(radius, nsample, xyz, new_xyz)
    got = F.query_ball_point(radius, nsample, torch.from_numpy(xyz).to(dev), torch.from_numpy(new_xyz).to(dev))
    assert got.dtype == torch.int64 and tuple(got.shape) == (B, S, nsample)
    assert np.array_equal(got.cpu().numpy(), ref)


def test_ball_query_multi_radius_one_scan(dev):
    x, xyz = _cloud(2, 2048, 21)
    st = make_start_idx(2, 2048, 1)
    new_xyz = R.index_points(xyz, R.farthest_point_sample(xyz, 512, st))
    radii, ks = [0.1, 0.2, 0.4], [32, 64, 128]
    t = torch.from_numpy(x).to(dev).transpose(1, 2)  # planar storage
    outs = F._ball_query_raw(radii, ks, t, torch.from_numpy(new_xyz).to(dev))
    for r, k, o in zip(radii, ks, outs):
        assert o.dtype == torch.int32
        assert np.array_equal(o.cpu().numpy().astype(np.int64), R.query_ball_point(r, k, xyz, new_xyz))


def test_ball_query_known_answers(dev):
    # hand-checkable from the source semantics (pointnet2_basic_layers.py:110-124)
    N = 100
    xyz = np.zeros((1, N, 3), np.float32)
    xyz[0, :, 0] = np.arange(N) * 0.01                       # points on a line, spacing 0.01
    q = xyz[:, [0, 50]].copy()
    # all in radius -> [0..K-1]
    got = F.query_ball_point(10.0, 8, torch.from_numpy(xyz).to(dev), torch.from_numpy(q).to(dev)).cpu().numpy()
    assert np.array_equal(got[0, 0], np.arange(8)) and np.array_equal(got[0, 1], np.arange(8))
    # only the query itself in radius -> K copies of its own index
    got = F.query_ball_point(0.001, 4, torch.from_numpy(xyz).to(dev), torch.from_numpy(q).to(dev)).cpu().numpy()
    assert np.array_equal(got[0, 0], [0, 0, 0, 0]) and np.array_equal(got[0, 1], [50, 50, 50, 50])
    # nobody in radius -> N everywhere (the source would then fail in index_points)
    far = np.full((1, 1, 3), 5.0, np.float32)
    got = F.query_ball_point(0.1, 4, torch.from_numpy(xyz).to(dev), torch.from_numpy(far).to(dev)).cpu().numpy()
    assert np.array_equal(got[0, 0], [N] * 4)
    # boundary: d == thr is INCLUDED (mask is strict >).  points at exact binary distances
    xyz2 = np.zeros((1, 4, 3), np.float32)
    xyz2[0, :, 0] = [0.0, 0.5, 1.0, 2.0]
    q2 = xyz2[:, [0]].copy()
    got = F.query_ball_point(0.5, 4, torch.from_numpy(xyz2).to(dev), torch.from_numpy(q2).to(dev)).cpu().numpy()
    assert np.array_equal(got[0, 0], [0, 1, 0, 0])         # thr = 0.25 == d(0, 0.5)


def test_square_distance_and_index_points(dev):
    x, xyz = _cloud(2, 500, 4)
    src = xyz[:, :70].copy()
    got = F.square_distance(torch.from_numpy(src).to(dev), torch.from_numpy(xyz).to(dev)).cpu().numpy()
    assert np.array_equal(got, R.square_distance(src, xyz))
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(2, 500, 37)).astype(np.float32)
    idx = rng.integers(0, 500, size=(2, 40, 6))
    for dt in (torch.int64, torch.int32, torch.float32):
        got = F.index_points(torch.from_numpy(pts).to(dev), torch.from_numpy(idx).to(dev).to(dt)).cpu().numpy()
        assert np.array_equal(got, R.index_points(pts, idx))


def test_index_points_backward(dev):
    rng = np.random.default_rng(3)
    pts = torch.from_numpy(rng.normal(size=(2, 50, 8)).astype(np.float32)).to(dev).requires_grad_(True)
    idx = torch.from_numpy(rng.integers(0, 50, size=(2, 30, 4))).to(dev)
    out = F.index_points(pts, idx)
    g = torch.randn_like(out)
    out.backward(g)
    ref = torch.zeros_like(pts)
    for b in range(2):
        ref[b].index_add_(0, idx[b].reshape(-1), g[b].reshape(-1, 8))
    assert torch.allclose(pts.grad, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("D", [0, 5, 128])
def test_sample_and_group(dev, D):
    x, xyz = _cloud(2, 1024, 8)
    st = make_start_idx(2, 1024, 2)
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(2, 1024, D)).astype(np.float32) if D else None
    ref_xyz, ref_pts = R.sample_and_group(128, 0.2, 32, xyz, pts, st)
    got_xyz, got_pts = F.sample_and_group(128, 0.2, 32, torch.from_numpy(xyz).to(dev),
                                          None if pts is None else torch.from_numpy(pts).to(dev),
                                          start_idx=torch.from_numpy(st).to(dev))
    assert np.array_equal(got_xyz.cpu().numpy(), ref_xyz)
    assert np.array_equal(got_pts.cpu().numpy(), ref_pts)   # gather + one fp32 subtraction: bit-exact


def test_full_size_properties(dev):
    """BASELINE config 2 sizes (B=32, N=4096): index-exact vs the C oracle + size-independent properties."""
    B, N, S, K = 32, 4096, 512, 32
    x, xyz = _cloud(B, N, 1234)
    st = make_start_idx(B, N, 1234)
    t = torch.from_numpy(x).to(dev).transpose(1, 2)
    idx, new_xyz = F._fps_raw(t, S, torch.from_numpy(st).to(dev))
    ref = R.farthest_point_sample(xyz, S, st)
    assert np.array_equal(idx.cpu().numpy(), ref)
    i = idx.cpu().numpy()
    assert all(len(set(row)) == S for row in i)                       # FPS never repeats a point on distinct clouds
    assert np.array_equal(i[:, 0], st)                                # first centroid is the start index
    g = F._ball_query_raw([0.2], [K], t, new_xyz)[0].cpu().numpy()
    assert np.array_equal(g.astype(np.int64), R.query_ball_point(0.2, K, xyz, R.index_points(xyz, ref)))
    assert (g >= 0).all() and (g < N).all()                           # queries are cloud points: always >= 1 hit
    assert (g[:, :, 0] <= i).all()                                    # the first hit is at most the query itself
    d = np.diff(g, axis=-1)
    # ascending until the padding starts, and padding repeats the first element
    for b in range(0, B, 7):
        for s in range(0, S, 97):
            row = g[b, s]
            n_unique = len(np.unique(row))
            assert (np.diff(row[:n_unique]) > 0).all() and (row[n_unique:] == row[0]).all()


@pytest.mark.parametrize("xyz_first", [True, False])
@pytest.mark.parametrize("D", [0, 5, 16])
def test_group_points_backward(dev, xyz_first, D):
    # papc_group_points_bwd_f32 (SURVEY 8b's group_gather backward) against torch's own gather autograd in float64
    rng = np.random.default_rng(3 + D)
    B, N, S, K = 2, 200, 24, 9
    xyz = torch.from_numpy(rng.normal(size=(B, N, 3)).astype(np.float32)).to(dev).requires_grad_()
    new_xyz = torch.from_numpy(rng.normal(size=(B, S, 3)).astype(np.float32)).to(dev).requires_grad_()
    feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_() if D else None
    idx = torch.from_numpy(rng.integers(0, N, size=(B, S, K)).astype(np.int32)).to(dev)
    idx[0, 0, :] = 7                                      # one neighbourhood made of copies of a single point
    w = torch.from_numpy(rng.normal(size=(B, S, K, 3 + D)).astype(np.float32)).to(dev)
    out = F.group_points(xyz, new_xyz, feats, idx, xyz_first=xyz_first)
    (out * w).sum().backward()
    x64, c64 = xyz.detach().double().requires_grad_(), new_xyz.detach().double().requires_grad_()
    f64 = feats.detach().double().requires_grad_() if D else None
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    gx = x64[bi, idx.long()] - c64[:, :, None, :]
    parts = [gx] + ([f64[bi, idx.long()]] if D else [])
    ref = torch.cat(parts if xyz_first else parts[::-1], -1)
    assert torch.equal(out.detach(), ref.float())
    (ref * w.double()).sum().backward()
    for got, want in ((xyz.grad, x64.grad), (new_xyz.grad, c64.grad)) + (((feats.grad, f64.grad),) if D else ()):
        assert (got.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
