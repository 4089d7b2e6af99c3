"""Seeded synthetic inputs shaped like the reference's data.

ShapeNet-part batches are ``[B,3,N]`` float32 clouds normalised like ``pc_normalize``
(/root/reference/PAPC/models/layers/pointnet2_basic_layers.py:17-23: subtract the centroid, divide
by the max norm) with ``[B,1]`` int64 labels (/root/reference/PAPC/datasets/pnloader.py:43-46).
KITTI-shaped pillar frames follow /root/reference/PAPC/models/detect/pointpillars/data/preprocess.py:236-244
(``voxels [P,T,4]`` f32, ``num_points [P]`` i32, ``coordinates [P,4]`` i32 = batch,z,y,x).
Pure numpy; used by tests, bench.py and the CPU baseline so every leg sees identical inputs.
"""
import numpy as np


def make_clouds(B, N, seed=1234):
    """[B,3,N] float32 clouds: unit-ball volume / sphere shell / box surface / 4-Gaussian mixture."""
    rng = np.random.default_rng(seed)
    out = np.empty((B, 3, N), np.float32)
    for b in range(B):
        kind = int(rng.integers(0, 4))
        if kind == 0:      # unit-ball volume
            v = rng.normal(size=(N, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            p = v * rng.random((N, 1)) ** (1.0 / 3.0)
        elif kind == 1:    # sphere shell
            v = rng.normal(size=(N, 3))
            p = v / np.linalg.norm(v, axis=1, keepdims=True)
        elif kind == 2:    # axis-aligned box surface
            p = rng.uniform(-1, 1, size=(N, 3)) * np.array([1.0, 0.6, 0.4])
            face = rng.integers(0, 3, size=N)
            sign = rng.choice([-1.0, 1.0], size=N)
            ext = np.array([1.0, 0.6, 0.4])
            p[np.arange(N), face] = sign * ext[face]
        else:              # 4-Gaussian mixture
            centers = rng.uniform(-0.6, 0.6, size=(4, 3))
            comp = rng.integers(0, 4, size=N)
            p = centers[comp] + rng.normal(scale=0.15, size=(N, 3))
        p = p.astype(np.float32)
        p = p - p.mean(axis=0, dtype=np.float32)
        m = np.max(np.sqrt(np.sum(p ** 2, axis=1))) or 1.0
        p = (p / m).astype(np.float32)
        out[b] = p.T
    return out


def make_labels(B, num_classes=16, seed=1234):
    rng = np.random.default_rng(seed + 7919)
    return rng.integers(0, num_classes, size=(B, 1)).astype(np.int64)


def make_start_idx(B, N, seed=1234):
    """Explicit FPS start indices (replaces paddle.randint, pointnet2_basic_layers.py:76)."""
    rng = np.random.default_rng(seed + 104729)
    return rng.integers(0, N, size=B).astype(np.int64)


def make_pillars(P=12000, T=100, seed=4321, nx=432, ny=496, voxel=(0.16, 0.16, 4.0),
                 pc_range=(0.0, -39.68, -3.0, 69.12, 39.68, 1.0)):
    """KITTI-shaped pillar frame: (voxels [P,T,4] f32, num_points [P] i32, coors [P,4] i32)."""
    rng = np.random.default_rng(seed)
    num_points = np.clip(rng.geometric(0.08, size=P), 1, T).astype(np.int32)
    cells = rng.choice(nx * ny, size=P, replace=False)
    cy, cx = cells // nx, cells % nx
    coors = np.stack([np.zeros(P, np.int64), np.zeros(P, np.int64), cy, cx], axis=1).astype(np.int32)
    voxels = np.zeros((P, T, 4), np.float32)
    u = rng.random((P, T, 4)).astype(np.float32)
    voxels[:, :, 0] = pc_range[0] + (cx[:, None] + u[:, :, 0]) * voxel[0]
    voxels[:, :, 1] = pc_range[1] + (cy[:, None] + u[:, :, 1]) * voxel[1]
    voxels[:, :, 2] = pc_range[2] + u[:, :, 2] * voxel[2]
    voxels[:, :, 3] = u[:, :, 3]
    mask = np.arange(T)[None, :] < num_points[:, None]
    voxels *= mask[:, :, None]
    return voxels, num_points, coors
