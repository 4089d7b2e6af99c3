"""Generates the committed golden vectors from the oracle (run from the repo root: python tests/golden/make_golden.py).

PARITY UNPINNED: the reference holds no golden vectors and cannot be executed here (PaddlePaddle absent), so these
fixtures are outputs of oracle/ (the source-following restatement), committed to pin the oracle and the HIP path
against each other and against drift.  Inputs are regenerated from papc_amd.synthetic with the stored seed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_np as R                      # noqa: E402
from papc_amd.synthetic import make_clouds, make_pillars, make_start_idx  # noqa: E402
from tests.util import seeded_weights                    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sampling(B, N, S, seed, name):
    xyz = np.ascontiguousarray(make_clouds(B, N, seed).transpose(0, 2, 1))
    st = make_start_idx(B, N, seed)
    fps = R.farthest_point_sample(xyz, S, st)
    out = dict(seed=seed, start_idx=st, fps_idx=fps, fps_idx_init1e10=R.farthest_point_sample(xyz, S, st, 1e10))
    new_xyz = R.index_points(xyz, fps)
    for r, k in [(0.1, 16), (0.2, 32), (0.4, 64), (0.8, 128)]:
        out["bq_r%s_k%d" % (str(r).replace(".", "p"), k)] = R.query_ball_point(r, k, xyz, new_xyz).astype(np.int32 if False else np.int64)
    np.savez_compressed(os.path.join(HERE, name), **out)


def sa_activations():
    B, N, seed = 2, 1024, 77
    x = make_clouds(B, N, seed)
    st = make_start_idx(B, N, seed)
    ws1 = seeded_weights([3, 64, 64, 128], 1)
    sa1 = R.PointNetSetAbstraction(128, 0.2, 32, 3, [64, 64, 128], False, ws1)
    l1_xyz, l1 = sa1.forward(x, None, st, f64=True)
    st2 = make_start_idx(B, 128, seed + 1)
    ws2 = seeded_weights([131, 128, 128, 256], 2)
    sa2 = R.PointNetSetAbstraction(32, 0.4, 64, 131, [128, 128, 256], False, ws2)
    l2_xyz, l2 = sa2.forward(l1_xyz, l1.astype(np.float32), st2, f64=True)
    np.savez_compressed(os.path.join(HERE, "sa_b2_n1024.npz"), seed=seed, start1=st, start2=st2,
                        l1_xyz=l1_xyz, l1_points=l1.astype(np.float32), l2_xyz=l2_xyz, l2_points=l2.astype(np.float32))


def pfn():
    voxels, nump, coors = make_pillars(P=64, T=100, seed=4321)
    nump[:4] = [1, 100, 2, 50]
    voxels *= (np.arange(100)[None, :] < nump[:, None])[:, :, None]
    rng = np.random.default_rng(9)
    w = (rng.normal(size=(64, 9)) * 0.3).astype(np.float32)
    g = rng.uniform(0.5, 1.5, 64).astype(np.float32) * rng.choice([1.0, 1.0, -1.0], 64).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    outs = {}
    for tag, vs, pr in [("ref_wiring", (1, 2, 3), (0, -40, -3, 70.4, 40, 1)), ("voxel_wiring", (0.16, 0.16, 4), (0, -39.68, -3, 69.12, 39.68, 1))]:
        outs["out_" + tag] = R.pillar_feature_net(voxels, nump, coors, [(w, g, b)], vs, pr, f64=True).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "pfn_p64.npz"), voxels=voxels, num_points=nump, coors=coors, w=w, gamma=g, beta=b, **outs)


def feature_propagation():
    """FP level 2 of the SSG segmentation net (in 384 -> [256,128]) at reduced size, both neighbour modes."""
    B, N, S, D1, D2, seed = 2, 256, 64, 16, 32, 99
    x1 = np.ascontiguousarray(make_clouds(B, N, seed))                       # [B,3,N]
    x2 = np.ascontiguousarray(x1[:, :, ::4])                                 # a subset, like FPS output
    rng = np.random.default_rng(seed)
    p1 = rng.normal(size=(B, D1, N)).astype(np.float32)
    p2 = rng.normal(size=(B, D2, S)).astype(np.float32)
    ws = seeded_weights([D1 + D2, 64, 32], 7)
    out = dict(seed=seed, points1=p1, points2=p2)
    for nb in ("reference", "nearest"):
        o, interp = R.PointNetFeaturePropagation(D1 + D2, [64, 32], ws, nb).forward(x1, x2, p1, p2, f64=True, return_interp=True)
        out["out_" + nb] = o.astype(np.float32)
        out["interp_" + nb] = interp
    d, i, w = R.three_nn_true(np.ascontiguousarray(x1.transpose(0, 2, 1)), np.ascontiguousarray(x2.transpose(0, 2, 1)))
    out.update(dist3=d, idx3=i.astype(np.int32), weight3=w)
    np.savez_compressed(os.path.join(HERE, "fp_b2_n256.npz"), **out)


if __name__ == "__main__":
    sampling(2, 1024, 128, 1234, "sampling_b2_n1024.npz")
    sampling(1, 4096, 512, 4242, "sampling_b1_n4096.npz")
    sa_activations()
    pfn()
    feature_propagation()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
