"""How much of the SA1 / SA2 neighbourhood lists is ball-query padding (copies of the first hit, pointnet2_basic_layers.py:118-124) on
SURFACE-sampled objects shaped like ShapeNet parts -- the data the reference trains on -- besides the benchmark's SURVEY 8d generator.
    gpurun -- 'python tools/probe/dup_fraction_surfaces.py'
Every cloud: N = 4096 points sampled uniformly by area on the object's surface, centred, scaled to the unit sphere (pc_normalize, :17-23);
SA1 = 512 FPS centroids, r = 0.2, nsample 32; SA2 = 128 of those 512, r = 0.4, nsample 64 (classify/pointnet2/pointnet2.py:11-15)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from papc_amd import functional as F
from papc_amd.synthetic import make_clouds


def _norm(p):
    p = p - p.mean(0)
    return (p / np.max(np.linalg.norm(p, axis=1))).astype(np.float32)


def _box(rng, n, ext, ctr=(0, 0, 0)):
    ext = np.asarray(ext, float)
    areas = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]])
    face = rng.choice(3, size=n, p=areas / areas.sum())
    p = rng.uniform(-1, 1, size=(n, 3)) * ext
    p[np.arange(n), face] = rng.choice([-1.0, 1.0], size=n) * ext[face]
    return p + np.asarray(ctr, float)


def _cyl(rng, n, r, h, ctr=(0, 0, 0), axis=2):
    t = rng.uniform(0, 2 * np.pi, n)
    z = rng.uniform(-h, h, n)
    p = np.stack([r * np.cos(t), r * np.sin(t), z], 1)
    if axis != 2:
        p = p[:, [2, 1, 0]] if axis == 0 else p[:, [0, 2, 1]]
    return p + np.asarray(ctr, float)


def _mix(rng, n, parts):
    """parts: (sampler, area weight); points split by area"""
    w = np.array([a for _, a in parts], float)
    cnt = rng.multinomial(n, w / w.sum())
    return np.concatenate([f(c) for (f, _), c in zip(parts, cnt) if c > 0], 0)


def shapes(rng, n):
    out = {}
    out["chair (seat, back, 4 legs)"] = _mix(rng, n, [
        (lambda c: _box(rng, c, (0.5, 0.5, 0.05)), 1.1), (lambda c: _box(rng, c, (0.5, 0.05, 0.5), (0, 0.45, 0.55)), 1.1),
        *[(lambda c, sx=sx, sy=sy: _cyl(rng, c, 0.04, 0.4, (0.42 * sx, 0.42 * sy, -0.45)), 0.2) for sx in (-1, 1) for sy in (-1, 1)]])
    out["airplane (fuselage, wings, tail)"] = _mix(rng, n, [
        (lambda c: _cyl(rng, c, 0.1, 1.0, axis=0), 1.26), (lambda c: _box(rng, c, (0.25, 0.9, 0.015)), 0.9),
        (lambda c: _box(rng, c, (0.1, 0.3, 0.01), (-0.9, 0, 0.02)), 0.12), (lambda c: _box(rng, c, (0.12, 0.01, 0.2), (-0.9, 0, 0.2)), 0.1)])
    out["table (top, 4 legs)"] = _mix(rng, n, [
        (lambda c: _box(rng, c, (0.9, 0.5, 0.04), (0, 0, 0.5)), 3.7),
        *[(lambda c, sx=sx, sy=sy: _cyl(rng, c, 0.04, 0.5, (0.8 * sx, 0.42 * sy, 0)), 0.25) for sx in (-1, 1) for sy in (-1, 1)]])
    out["mug (cylinder wall + bottom)"] = _mix(rng, n, [(lambda c: _cyl(rng, c, 0.5, 0.6), 3.8), (lambda c: _box(rng, c, (0.35, 0.35, 0.001), (0, 0, -0.6)), 0.5)])
    v = rng.normal(size=(n, 3))
    out["sphere shell"] = v / np.linalg.norm(v, axis=1, keepdims=True)
    out["lamp (thin pole, shade, base)"] = _mix(rng, n, [
        (lambda c: _cyl(rng, c, 0.02, 0.8), 0.2), (lambda c: _cyl(rng, c, 0.35, 0.2, (0, 0, 0.8)), 0.9), (lambda c: _cyl(rng, c, 0.3, 0.02, (0, 0, -0.8)), 0.1)])
    return {k: _norm(p) for k, p in out.items()}


def stats(x, dev):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)                     # [B, N, 3]
    B, N = t.shape[:2]
    st = torch.zeros(B, dtype=torch.int64, device=dev)
    _, nx1 = F._fps_raw(t, 512, st)
    i1 = F._ball_query_raw([0.2], [32], t, nx1)[0]
    _, nx2 = F._fps_raw(nx1, 128, st)
    i2 = F._ball_query_raw([0.4], [64], nx1, nx2)[0]
    res = []
    for idx in (i1, i2):
        dup = idx == idx[:, :, :1]
        dup[:, :, 0] = False
        K = idx.shape[2]
        uniq = K - dup.sum(2)
        c8 = (uniq + 7) // 8 * 8
        res.append((100.0 * dup.float().mean().item(), uniq.float().mean().item(), 100.0 * (uniq == K).float().mean().item(),
                    c8.float().sum().item() / (idx.shape[0] * idx.shape[1] * K)))
    return res


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2024)
    B = 16
    per = {}
    for _ in range(B):
        for k, p in shapes(rng, 4096).items():
            per.setdefault(k, []).append(p)
    print("%-36s | SA1 r=0.2 K=32: padding %%, distinct/group, full %% | SA2 r=0.4 K=64: padding %%, distinct/group, full %%, compacted rows / padded rows" % "object")
    for k, ps in per.items():
        (d1, u1, f1, _), (d2, u2, f2, c2) = stats(np.stack(ps), dev)
        print("%-36s | %5.1f %5.1f %5.1f | %5.1f %5.1f %5.1f %5.3f" % (k, d1, u1, f1, d2, u2, f2, c2))
    x = make_clouds(32, 4096, 1234).transpose(0, 2, 1)
    (d1, u1, f1, _), (d2, u2, f2, c2) = stats(x, dev)
    print("%-36s | %5.1f %5.1f %5.1f | %5.1f %5.1f %5.1f %5.3f" % ("bench.py generator (SURVEY 8d)", d1, u1, f1, d2, u2, f2, c2))
