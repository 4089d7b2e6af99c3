"""GPU parity of the voxeliser and PointPillarsScatter: bit-exact against the oracle's literal loop."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R
from papc_amd.voxel import PointPillarsScatter, points_to_voxel

pytestmark = pytest.mark.gpu

KITTI = dict(voxel_size=(0.16, 0.16, 4.0), coors_range=(0, -39.68, -3, 69.12, 39.68, 1))


def _frame(n, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 4), np.float32)
    pts[:, 0] = rng.uniform(-5, 75, n) * spread
    pts[:, 1] = rng.uniform(-45, 45, n) * spread
    pts[:, 2] = rng.uniform(-4, 2, n)
    pts[:, 3] = rng.uniform(0, 1, n)
    # clusters: many points per pillar near the sensor
    k = n // 3
    pts[:k, 0] = rng.normal(10, 1.0, k); pts[:k, 1] = rng.normal(0, 1.0, k)
    rng.shuffle(pts)
    return pts


@pytest.mark.parametrize("n,max_points,max_voxels,reverse", [
    (3000, 35, 20000, True),      # nothing clipped
    (3000, 5, 20000, False),      # max_points clips the dense pillars; xyz coordinate order
    (4000, 100, 150, True),       # max_voxels reached: the source's `break` drops every later point
    (200, 100, 12000, True),
])
def test_points_to_voxel_bit_exact(dev, n, max_points, max_voxels, reverse):
    pts = _frame(n, 7 + n)
    rv, rc, rn = R.points_to_voxel(pts, KITTI["voxel_size"], KITTI["coors_range"], max_points, reverse, max_voxels)
    v, c, m = points_to_voxel(torch.from_numpy(pts).to(dev), KITTI["voxel_size"], KITTI["coors_range"], max_points, reverse, max_voxels)
    assert v.shape == rv.shape and c.shape == rc.shape
    assert np.array_equal(c.cpu().numpy(), rc)
    assert np.array_equal(m.cpu().numpy(), rn)
    assert np.array_equal(v.cpu().numpy(), rv)


def test_points_to_voxel_edges(dev):
    # every point outside the range -> no voxels; points exactly on the upper bound are dropped (c >= grid_size)
    pts = np.array([[100.0, 0, 0, 0], [69.12, 0, 0, 0], [0.0, -39.68, -3.0, 0.5], [0.159, -39.68, -3.0, 0.7]], np.float32)
    v, c, m = points_to_voxel(torch.from_numpy(pts).to(dev), **KITTI, max_points=4, max_voxels=10)
    rv, rc, rn = R.points_to_voxel(pts, **KITTI, max_points=4, max_voxels=10)
    assert np.array_equal(c.cpu().numpy(), rc) and np.array_equal(m.cpu().numpy(), rn) and np.array_equal(v.cpu().numpy(), rv)
    assert m.cpu().numpy().tolist() == [2]
    pv, pc, pn, cnt = points_to_voxel(torch.from_numpy(pts).to(dev), **KITTI, max_points=4, max_voxels=10, padded=True)
    assert int(cnt) == 1 and pv.shape == (10, 4, 4) and not pv[1:].any()


def test_full_size_frame_properties(dev):
    """KITTI-sized frame (120k points, 12000 x 100): too slow for the Python loop in seconds, so invariants: counts
    match a histogram of the cell ids, every stored point lies in its pillar, pillars are numbered by first appearance."""
    n = 120000
    pts = _frame(n, 3)
    t = torch.from_numpy(pts).to(dev)
    v, c, m = points_to_voxel(t, **KITTI, max_points=100, max_voxels=12000)
    v, c, m = v.cpu().numpy(), c.cpu().numpy(), m.cpu().numpy()
    assert len(m) <= 12000 and m.min() >= 1 and m.max() <= 100
    vs, lo = np.array(KITTI["voxel_size"], np.float32), np.array(KITTI["coors_range"][:3], np.float32)
    for p in range(0, len(m), 97):
        cell = np.floor((v[p, :m[p], :3] - lo) / vs).astype(np.int32)
        assert (cell[:, ::-1] == c[p]).all()
        assert not v[p, m[p]:].any()
    assert len({tuple(r) for r in c}) == len(c)            # one pillar per cell
    # first-appearance order: the first point of pillar p appears in the frame before the first point of pillar p+1
    first = {}
    cells_all = np.floor((pts[:, :3] - lo) / vs).astype(np.int64)
    grid = np.array([432, 496, 1])
    ok = ((cells_all >= 0) & (cells_all < grid)).all(1)
    keys = cells_all[:, 2] * 10**8 + cells_all[:, 1] * 10**4 + cells_all[:, 0]
    for i in np.nonzero(ok)[0]:
        first.setdefault(int(keys[i]), int(i))
    order = [first[int(r[0]) * 10**8 + int(r[1]) * 10**4 + int(r[2])] for r in c[:2000]]
    assert order == sorted(order)


def test_pillar_scatter_forward_backward(dev):
    rng = np.random.default_rng(1)
    P, C, B, ny, nx = 500, 64, 2, 31, 27
    coords = np.zeros((P, 4), np.int32)
    coords[:, 0] = rng.integers(0, B, P)
    cells = rng.permutation(ny * nx)[:P // 2]
    coords[:P // 2, 2], coords[:P // 2, 3] = cells // nx, cells % nx
    coords[P // 2:, 2] = rng.integers(0, ny, P - P // 2)      # the second half collides with the first: last pillar wins
    coords[P // 2:, 3] = rng.integers(0, nx, P - P // 2)
    feats = rng.normal(size=(P, C)).astype(np.float32)
    ref = R.pillar_scatter(feats, coords, B, ny, nx)
    tf = torch.from_numpy(feats).to(dev).requires_grad_(True)
    sc = PointPillarsScatter([1, C, ny, nx], num_input_features=C)
    out = sc(tf, torch.from_numpy(coords).to(dev), B)
    assert np.array_equal(out.detach().cpu().numpy(), ref)
    g = torch.from_numpy(rng.normal(size=ref.shape).astype(np.float32)).to(dev)
    out.backward(g)
    # reference gradient: only the winning pillar of a cell receives it
    gref = np.zeros_like(feats)
    owner = -np.ones((B, ny, nx), np.int64)
    for p in range(P):
        owner[coords[p, 0], coords[p, 2], coords[p, 3]] = p
    gn = g.cpu().numpy()
    for p in range(P):
        b, y, x = coords[p, 0], coords[p, 2], coords[p, 3]
        if owner[b, y, x] == p:
            gref[p] = gn[b, :, y, x]
    assert np.array_equal(tf.grad.cpu().numpy(), gref)


def test_frame_to_bev_pipeline(dev):
    """points -> points_to_voxel -> PillarFeatureNet -> PointPillarsScatter on the device (BASELINE config 5 shapes, 2 frames):
    the BEV canvas holds exactly the PFN output of every pillar at its cell, zeros elsewhere; gradients reach the PFN."""
    from papc_amd.pillars import PillarFeatureNet
    vox, coords, nums = [], [], []
    for b in range(2):
        v, c, m = points_to_voxel(torch.from_numpy(_frame(60000, 11 + b)).to(dev), **KITTI, max_points=100, max_voxels=12000)
        vox.append(v); nums.append(m)
        coords.append(torch.cat([torch.full((len(c), 1), b, device=dev, dtype=torch.int32), c], 1))   # (batch, z, y, x)
    vox, coords, nums = torch.cat(vox), torch.cat(coords), torch.cat(nums)
    pfn = PillarFeatureNet(num_filters=(64,), voxel_size=KITTI["voxel_size"], pc_range=KITTI["coors_range"]).to(dev)
    feats = pfn(vox, nums, coords)
    assert feats.shape == (len(vox), 64)
    bev = PointPillarsScatter([1, 64, 496, 432], num_input_features=64)(feats, coords, 2)
    assert bev.shape == (2, 64, 496, 432)
    p = len(vox) // 3
    b, y, x = (int(t) for t in (coords[p, 0], coords[p, 2], coords[p, 3]))
    assert torch.equal(bev[b, :, y, x], feats[p])
    assert int((bev.abs().sum(1) > 0).sum()) <= len(vox)
    bev.square().mean().backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in pfn.parameters())
