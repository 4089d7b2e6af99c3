"""CPU oracle for the PAPC hot path -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``papc_amd`` never does; it fails loudly if its HIP library is missing.

PARITY UNPINNED.  The reference (AgentMaker/PAPC) is pure Python on PaddlePaddle; it holds no
tests, fixtures or golden vectors for this path, and PaddlePaddle cannot be installed in the build
container, so this restatement could not be checked against an execution of the reference.  It
follows the reference source line by line (``file:line`` citations are into
``/root/reference/PAPC/models/layers/pointnet2_basic_layers.py`` unless another file is named),
numpy standing in for paddle, with fp32 rounding order fixed to the canonical arithmetic of
SURVEY.md section 8a (see ``papc_oracle.c``).  Second opinions available here: the same functions
written literally (``*_literal`` below: tile/mask/sort exactly as the source does) and a torch-CPU
transliteration (``torch_cpu_reference.py``); the tests check all three agree.

Layout conventions are the reference's: xyz ``[B,N,3]``, features ``[B,N,D]`` inside the free
functions; layers take ``[B,C,N]``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """Load (building with gcc if needed) the C restatement ``liboracle.so``."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "papc_oracle.c")
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = ctypes.CDLL(so)
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------------------------
# square_distance :26-40
# --------------------------------------------------------------------------------------------
def _matmul_nt(a, b):
    """paddle.matmul(a, b.transpose([0,2,1])) with the canonical k-ordered fma chain (C)."""
    a, b = _f32(a), _f32(b)
    B, N, C = a.shape
    M = b.shape[1]
    out = np.empty((B, N, M), np.float32)
    _lib().orc_matmul_nt(_fp(a), _fp(b), B, N, M, C, _fp(out))
    return out


def square_distance(src, dst):
    """:26-40, written as the source writes it (three in-place steps, fp32)."""
    src, dst = _f32(src), _f32(dst)
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = np.float32(-2) * _matmul_nt(src, dst)                                   # :36
    dist += _sum_last3(src ** 2).reshape(B, N, 1)                                  # :37
    dist += _sum_last3(dst ** 2).reshape(B, 1, M)                                  # :38
    return dist


def _sum_last3(sq):
    """paddle.sum(x, axis=-1) over a length-3 axis: canonical left-to-right (a0+a1)+a2."""
    assert sq.shape[-1] == 3
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def square_distance_c(src, dst):
    """Same values from the C restatement (one fused loop); used for big inputs."""
    src, dst = _f32(src), _f32(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = np.empty((B, N, M), np.float32)
    _lib().orc_square_distance(_fp(src), _fp(dst), B, N, M, _fp(out))
    return out


# --------------------------------------------------------------------------------------------
# index_points :43-62
# --------------------------------------------------------------------------------------------
def index_points(points, idx):
    """:43-62 -- numpy fancy indexing exactly as the source (idx may be float, cast to int64 :59)."""
    points = np.asarray(points)
    B = points.shape[0]
    idx = np.asarray(idx)
    view_shape = list(idx.shape)
    view_shape[1:] = [1] * (len(view_shape) - 1)
    repeat_shape = list(idx.shape)
    repeat_shape[0] = 1
    batch_indices = np.tile(np.arange(B).reshape(view_shape), repeat_shape)         # :56
    return points[batch_indices.astype("int64"), idx.astype("int64"), :]            # :57-60


# --------------------------------------------------------------------------------------------
# farthest_point_sample :65-95
# --------------------------------------------------------------------------------------------
def farthest_point_sample_literal(xyz, npoint, start_idx, init_dist=1.0):
    """:65-95 step by step.  ``start_idx`` replaces paddle.randint (:76).  Returns float32
    centroids like the source (:74)."""
    xyz = _f32(xyz)
    B, N, C = xyz.shape
    centroids = np.zeros((B, npoint), np.float32)                                    # :74
    distance = np.full((B, N), init_dist, np.float32)                               # :75 (ones)
    farthest = np.asarray(start_idx, dtype=np.int64).copy()                         # :76
    batch_indices = np.arange(B)
    for i in range(npoint):                                                         # :79
        centroids[:, i] = farthest                                                  # :80
        centroid = xyz[batch_indices, farthest, :][:, None, :]                      # :81-85
        dist = _sum_last3((xyz - centroid) ** 2)                                    # :86
        mask = dist < distance                                                      # :87
        distance[mask] = dist[mask]                                                 # :88-92
        farthest = np.argmax(distance, -1)                                          # :93 (first max)
    return centroids


def farthest_point_sample(xyz, npoint, start_idx, init_dist=1.0):
    """C restatement of the same loop; returns int32 indices ``[B,npoint]``."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    start = np.ascontiguousarray(start_idx, dtype=np.int64)
    out = np.empty((B, npoint), np.int32)
    _lib().orc_fps(_fp(xyz), B, N, npoint, _fp(start), ctypes.c_float(init_dist), _fp(out))
    return out


# --------------------------------------------------------------------------------------------
# query_ball_point :98-126
# --------------------------------------------------------------------------------------------
def radius_threshold(radius):
    """``radius ** 2`` is a python double; the comparison promotes it to the tensor dtype (:112)."""
    return np.float32(float(radius) * float(radius))


def query_ball_point_literal(radius, nsample, xyz, new_xyz):
    """:98-126 exactly as written: tile arange, mask -> N, full sort, keep nsample, pad with first."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, C = xyz.shape
    _, S, _ = new_xyz.shape
    group_idx = np.tile(np.arange(N, dtype=np.int64).reshape(1, 1, N), [B, S, 1])   # :110
    sqrdists = square_distance(new_xyz, xyz)                                        # :111
    mask = sqrdists > radius_threshold(radius)                                      # :112
    group_idx[mask] = N                                                             # :113-116
    group_idx = np.sort(group_idx, axis=-1)[:, :, :nsample]                         # :117
    group_first = np.tile(group_idx[:, :, 0].reshape(B, S, 1), [1, 1, nsample])     # :118
    mask = group_idx == N                                                           # :119
    group_idx[mask] = group_first[mask]                                             # :120-123
    return group_idx


def query_ball_point(radius, nsample, xyz, new_xyz):
    """C restatement (first-``nsample``-hits scan); int64 ``[B,S,nsample]``."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), np.int64)
    _lib().orc_ball_query(_fp(xyz), _fp(new_xyz), B, N, S, ctypes.c_float(radius_threshold(radius)),
                          nsample, _fp(out))
    return out


# --------------------------------------------------------------------------------------------
# sample_and_group :129-157, sample_and_group_all :160-176
# --------------------------------------------------------------------------------------------
def sample_and_group(npoint, radius, nsample, xyz, points, start_idx, returnfps=False, init_dist=1.0):
    xyz = _f32(xyz)
    B, N, C = xyz.shape
    S = npoint
    fps_idx = farthest_point_sample(xyz, npoint, start_idx, init_dist)              # :143
    new_xyz = index_points(xyz, fps_idx)                                            # :144
    idx = query_ball_point(radius, nsample, xyz, new_xyz)                           # :145
    grouped_xyz = index_points(xyz, idx)                                            # :146
    grouped_xyz_norm = grouped_xyz - new_xyz.reshape(B, S, 1, C)                    # :147
    if points is not None:
        grouped_points = index_points(_f32(points), idx)                            # :150
        new_points = np.concatenate([grouped_xyz_norm, grouped_points], axis=-1)    # :151 xyz first
    else:
        new_points = grouped_xyz_norm                                               # :153
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    xyz = _f32(xyz)
    B, N, C = xyz.shape
    new_xyz = np.zeros((B, 1, C), np.float32)                                       # :170
    grouped_xyz = xyz.reshape(B, 1, N, C)                                           # :171
    if points is not None:
        new_points = np.concatenate([grouped_xyz, _f32(points).reshape(B, 1, N, -1)], axis=-1)  # :173
    else:
        new_points = grouped_xyz
    return new_xyz, new_points


# --------------------------------------------------------------------------------------------
# Conv2D(1x1) + BatchNorm2D(train) + ReLU stack and max  :215-219
# --------------------------------------------------------------------------------------------
def conv1x1_rows(x, w, bias, f64=False):
    """nn.Conv2D(cin, cout, 1) on rows: x [M,Cin] . w[Cout,Cin]^T + bias (:189, :217).
    fp32 path = canonical k-ordered fmaf chain (C); f64 path = numpy double (error bound)."""
    if f64:
        y = np.asarray(x, np.float64) @ np.asarray(w, np.float64).T
        return y + np.asarray(bias, np.float64) if bias is not None else y
    x, w = _f32(x), _f32(w)
    M, Cin = x.shape
    Cout = w.shape[0]
    y = np.empty((M, Cout), np.float32)
    b32 = _f32(bias) if bias is not None else None
    _lib().orc_conv1x1(_fp(x), _fp(w), _fp(b32) if b32 is not None else None,
                       ctypes.c_int64(M), Cin, Cout, _fp(y))
    return y


def batchnorm_train_rows(y, gamma, beta, eps, f64=False):
    """BatchNorm (training): per-channel batch mean / biased variance over all rows, then
    (y-mean)/sqrt(var+eps)*gamma+beta (:190, :217; paddle default eps 1e-5).  Statistics are
    accumulated in double in both modes; returns (out, mean, var)."""
    yd = np.asarray(y, np.float64)
    mean = yd.mean(axis=0)
    var = yd.var(axis=0)
    if f64:
        out = (yd - mean) / np.sqrt(var + eps) * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
        return out, mean, var
    mean32, var32 = mean.astype(np.float32), var.astype(np.float32)
    inv = (np.float32(1.0) / np.sqrt(var32 + np.float32(eps))).astype(np.float32)
    out = (np.asarray(y, np.float32) - mean32) * inv * _f32(gamma) + _f32(beta)
    return out.astype(np.float32), mean, var


def mlp_stack_rows(x, weights, f64=False, eps=1e-5, return_all=False):
    """relu(bn(conv(x))) for each (w, bias, gamma, beta) in ``weights`` on rows [M,C] (:215-217)."""
    acts = []
    for (w, b, g, bt) in weights:
        y = conv1x1_rows(x, w, b, f64)
        z, _, _ = batchnorm_train_rows(y, g, bt, eps, f64)
        x = np.maximum(z, 0)
        acts.append((y, x))
    return (x, acts) if return_all else x


class PointNetSetAbstraction:
    """:179-221.  ``weights`` = list of (conv_w [Cout,Cin], conv_b [Cout], bn_gamma, bn_beta)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, weights):
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.weights = weights
        assert len(weights) == len(mlp)

    def forward(self, xyz, points, start_idx=None, f64=False, init_dist=1.0, return_all=False):
        xyz = np.transpose(_f32(xyz), (0, 2, 1))                                    # :203
        if points is not None:
            points = np.transpose(_f32(points), (0, 2, 1))                          # :205
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)                 # :211
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points,
                                                   start_idx, init_dist=init_dist)  # :213
        B, S, K, C = new_points.shape
        # :214-217: [B,C,K,S] conv/bn/relu == the same ops on rows (b,s,k) x C
        rows = new_points.reshape(B * S * K, C)
        z, acts = mlp_stack_rows(rows, self.weights, f64, return_all=True)
        z = z.reshape(B, S, K, -1)
        out = z.max(axis=2)                                                         # :219 max over nsample
        out = np.transpose(out, (0, 2, 1))                                          # [B,D',S]
        new_xyz = np.transpose(new_xyz, (0, 2, 1))                                  # :220
        if return_all:
            return new_xyz, out, acts
        return new_xyz, out


class PointNetSetAbstractionMsg:
    """:224-281.  ``weights[i]`` is the (w,b,gamma,beta) list of radius branch i."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, weights):
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.weights = weights

    def forward(self, xyz, points, start_idx, f64=False, init_dist=1.0):
        xyz = np.transpose(_f32(xyz), (0, 2, 1))                                    # :252
        if points is not None:
            points = np.transpose(_f32(points), (0, 2, 1))
        B, N, C = xyz.shape
        S = self.npoint
        new_xyz = index_points(xyz, farthest_point_sample(xyz, S, start_idx, init_dist))   # :258
        outs = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            group_idx = query_ball_point(radius, K, xyz, new_xyz)                   # :262
            grouped_xyz = index_points(xyz, group_idx)
            grouped_xyz = grouped_xyz - new_xyz.reshape(B, S, 1, C)                 # :264
            if points is not None:
                grouped_points = index_points(points, group_idx)
                grouped_points = np.concatenate([grouped_points, grouped_xyz], axis=-1)    # :267 feats first
            else:
                grouped_points = grouped_xyz
            rows = grouped_points.reshape(B * S * K, -1)
            z = mlp_stack_rows(rows, self.weights[i], f64).reshape(B, S, K, -1)
            outs.append(np.transpose(z.max(axis=2), (0, 2, 1)))                     # :276
        return np.transpose(new_xyz, (0, 2, 1)), np.concatenate(outs, axis=1)       # :279-280


# --------------------------------------------------------------------------------------------
# PointNetFeaturePropagation  :284-335
# --------------------------------------------------------------------------------------------
def three_nn_literal(xyz1, xyz2):
    """:315-322 exactly as written: sort the distance matrix, THEN argsort the sorted matrix (so idx is the
    argsort of an ascending row: 0,1,2 whenever the three smallest distances are distinct), keep 3, inverse-distance
    weights.  Returns (dists [B,N,3], idx [B,N,3] int64, weight [B,N,3]).  ``kind='stable'`` pins tie order (paddle's
    argsort leaves it unspecified)."""
    dists = square_distance_c(xyz1, xyz2)                                           # :315
    dists = np.sort(dists, axis=-1)                                                 # :316
    idx = np.argsort(dists, axis=-1, kind="stable")                                 # :317 (of the SORTED matrix)
    dists, idx = dists[:, :, :3], idx[:, :, :3]                                     # :318
    dist_recip = (np.float32(1.0) / (dists + np.float32(1e-8))).astype(np.float32)  # :320
    norm = (dist_recip[:, :, 0:1] + dist_recip[:, :, 1:2]) + dist_recip[:, :, 2:3]  # :321 sum over 3, left to right
    weight = (dist_recip / norm).astype(np.float32)                                 # :322
    return dists, idx.astype(np.int64), weight


def three_nn_true(xyz1, xyz2):
    """The neighbours the paper intends (what :316-317 would give with the two lines swapped): stable argsort of
    the unsorted matrix.  Same distances / weights as the literal version, real indices."""
    d = square_distance_c(xyz1, xyz2)
    idx = np.argsort(d, axis=-1, kind="stable")[:, :, :3]
    dists, _, weight = three_nn_literal(xyz1, xyz2)
    return dists, idx.astype(np.int64), weight


def three_interpolate(points2, idx, weight):
    """:323  sum(index_points(points2, idx) * weight.reshape(B,N,3,1), axis=2) -- products rounded to fp32, then
    added left to right."""
    B, N, _ = idx.shape
    g = index_points(_f32(points2), idx)                                            # [B,N,3,D]
    prod = (g * _f32(weight).reshape(B, N, 3, 1)).astype(np.float32)
    return ((prod[:, :, 0] + prod[:, :, 1]) + prod[:, :, 2]).astype(np.float32)


class PointNetFeaturePropagation:
    """:284-335.  ``weights`` = list of (conv_w [Cout,Cin], conv_b, bn_gamma, bn_beta) (Conv1D k=1 + BatchNorm1D)."""

    def __init__(self, in_channel, mlp, weights, neighbours="reference"):
        self.weights = weights
        self.neighbours = neighbours
        assert len(weights) == len(mlp)

    def forward(self, xyz1, xyz2, points1, points2, f64=False, return_interp=False):
        xyz1 = np.transpose(_f32(xyz1), (0, 2, 1))                                  # :305
        xyz2 = np.transpose(_f32(xyz2), (0, 2, 1))                                  # :306
        points2 = np.transpose(_f32(points2), (0, 2, 1))                            # :308
        B, N, C = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interpolated = np.tile(points2, (1, N, 1))                              # :312
        else:
            fn = three_nn_literal if self.neighbours == "reference" else three_nn_true
            _, idx, weight = fn(xyz1, xyz2)
            interpolated = three_interpolate(points2, idx, weight)                  # :323
        if points1 is not None:
            points1 = np.transpose(_f32(points1), (0, 2, 1))                        # :326
            new_points = np.concatenate([points1, interpolated], axis=-1)           # :327
        else:
            new_points = interpolated
        # :331-333 Conv1D/BatchNorm1D/relu on [B,D,N] == the same ops on rows (b,n) x D
        rows = new_points.reshape(B * N, -1)
        z = mlp_stack_rows(rows, self.weights, f64)
        out = np.transpose(z.reshape(B, N, -1), (0, 2, 1))
        return (out, interpolated) if return_interp else out


# --------------------------------------------------------------------------------------------
# PointPillars PFN: /root/reference/PAPC/models/detect/pointpillars/models/bones/pillars.py
# --------------------------------------------------------------------------------------------
def get_paddings_indicator(actual_num, max_num):
    """libs/tools/__init__.py:26-35 (axis=0): mask[p,t] = actual_num[p] > t."""
    return np.asarray(actual_num).astype(np.int64)[:, None] > np.arange(max_num, dtype=np.int64)[None, :]


def pillar_decorate(features, num_voxels, coors, vx, vy, x_offset, y_offset, with_distance=False):
    """pillars.py:79-102 -> masked 9-channel rows [P,T,9] (fp32, op order as written); with_distance (:92-94) appends
    paddle.norm(features[:, :, :3], 2, 2, keepdim=True) as a 10th channel."""
    features = _f32(features)
    nv = np.asarray(num_voxels).astype(np.float32).reshape(-1, 1, 1)
    points_mean = features[:, :, :3].sum(axis=1, keepdims=True, dtype=np.float32) / nv      # :82
    f_cluster = features[:, :, :3] - points_mean                                             # :83
    f_center = np.zeros_like(features[:, :, :2])                                             # :86
    cx = np.asarray(coors)[:, 3].astype(np.float32)[:, None]
    cy = np.asarray(coors)[:, 2].astype(np.float32)[:, None]
    f_center[:, :, 0] = features[:, :, 0] - (cx * np.float32(vx) + np.float32(x_offset))     # :87
    f_center[:, :, 1] = features[:, :, 1] - (cy * np.float32(vy) + np.float32(y_offset))     # :88
    ls = [features, f_cluster, f_center]                                                     # :91
    if with_distance:                                                                        # :92-94
        sq = features[:, :, :3] * features[:, :, :3]
        ls.append(np.sqrt((sq[:, :, 0:1] + sq[:, :, 1:2]) + sq[:, :, 2:3]).astype(np.float32))
    feats = np.concatenate(ls, axis=-1)                                                      # :95
    mask = get_paddings_indicator(num_voxels, feats.shape[1])                                # :99-100
    feats = feats * mask[..., None].astype(np.float32)                                       # :101-102
    return feats


def pfn_layer(inputs, w, gamma, beta, eps=1e-3, last_layer=True, f64=False):
    """pillars.py:29-41.  ``w`` is [Cout,Cin] (paddle Linear stores [in,out]; pass the transpose)."""
    P, T, Cin = inputs.shape
    y = conv1x1_rows(inputs.reshape(P * T, Cin), w, None, f64)                               # :30
    z, _, _ = batchnorm_train_rows(y, gamma, beta, eps, f64)                                 # :31
    x = np.maximum(z, 0).reshape(P, T, -1)                                                   # :32
    x_max = x.max(axis=1, keepdims=True)                                                     # :34
    if last_layer:
        return x_max                                                                         # :36-37
    return np.concatenate([x, np.tile(x_max, (1, T, 1))], axis=2)                            # :39-41


def pillar_feature_net(features, num_voxels, coors, layer_weights, voxel_size=(0.2, 0.2, 4),
                       pc_range=(0, -40, -3, 70.4, 40, 1), f64=False):
    """pillars.py:43-108.  ``layer_weights`` = [(w, gamma, beta), ...] one per PFNLayer."""
    vx, vy = voxel_size[0], voxel_size[1]
    x_offset = vx / 2 + pc_range[0]                                                          # :76
    y_offset = vy / 2 + pc_range[1]                                                          # :77
    feats = pillar_decorate(features, num_voxels, coors, vx, vy, x_offset, y_offset)
    n = len(layer_weights)
    for i, (w, g, b) in enumerate(layer_weights):
        feats = pfn_layer(feats, w, g, b, last_layer=(i == n - 1), f64=f64)                  # :105-106
    return np.squeeze(feats)                                                                 # :108


# --------------------------------------------------------------------------------------------
# points_to_voxel  (pointpillars/libs/ops/point_cloud/point_cloud_ops.py) and PointPillarsScatter (bones/pillars.py:110-142)
# --------------------------------------------------------------------------------------------
def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """:106-166 with the loop of :8-53 (reverse) / :56-103 written out: sequential first-come assignment, float32
    arithmetic like the numba-jitted source (its arrays are float32)."""
    points = _f32(points)
    voxel_size = np.asarray(voxel_size, dtype=points.dtype)                         # :137-138
    coors_range = np.asarray(coors_range, dtype=points.dtype)
    grid_size = np.round((coors_range[3:] - coors_range[:3]) / voxel_size).astype(np.int32)   # :25 / :140-141
    shape = tuple(grid_size.tolist())
    if reverse_index:
        shape = shape[::-1]                                                         # :142-143
    num_points_per_voxel = np.zeros((max_voxels,), np.int32)
    coor_to_voxelidx = -np.ones(shape, np.int32)
    voxels = np.zeros((max_voxels, max_points, points.shape[-1]), points.dtype)
    coors = np.zeros((max_voxels, 3), np.int32)
    coor = np.zeros((3,), np.int32)
    voxel_num = 0
    for i in range(points.shape[0]):
        failed = False
        for j in range(3):
            c = np.floor((points[i, j] - coors_range[j]) / voxel_size[j])           # :35 (float32)
            if c < 0 or c >= grid_size[j]:
                failed = True
                break
            coor[2 - j if reverse_index else j] = c                                 # :39 / :87
        if failed:
            continue
        voxelidx = coor_to_voxelidx[coor[0], coor[1], coor[2]]
        if voxelidx == -1:
            voxelidx = voxel_num
            if voxel_num >= max_voxels:
                break                                                               # :44-45
            voxel_num += 1
            coor_to_voxelidx[coor[0], coor[1], coor[2]] = voxelidx
            coors[voxelidx] = coor
        num = num_points_per_voxel[voxelidx]
        if num < max_points:                                                        # :48-50
            voxels[voxelidx, num] = points[i]
            num_points_per_voxel[voxelidx] += 1
    return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


def pillar_scatter(voxel_features, coords, batch_size, ny, nx):
    """PointPillarsScatter.forward, pillars.py:122-142 (select_change = numpy fancy assignment: last duplicate wins)."""
    voxel_features = _f32(voxel_features)
    C = voxel_features.shape[1]
    out = []
    for b in range(batch_size):
        canvas = np.zeros((C, nx * ny), np.float32)
        mask = coords[:, 0] == b
        if mask.any():
            this = coords[mask]
            indices = (this[:, 2] * nx + this[:, 3]).astype(np.int64)               # :131-132
            canvas[:, indices] = voxel_features[mask].T                             # :133-136
        out.append(canvas)
    return np.stack(out, 0).reshape(batch_size, C, ny, nx)


# ---------------------------------------------------------------------------------------------------------------------
# axis-aligned bitmask NMS
# ---------------------------------------------------------------------------------------------------------------------
def nms_iou(a, b):
    """iou_device, non_max_suppression/nms_gpu.py:22-34: fp32, boxes are inclusive pixel ranges (the "+1")."""
    f = np.float32
    one, zero = f(1.0), f(0.0)
    left, right = max(a[0], b[0]), min(a[2], b[2])
    top, bottom = max(a[1], b[1]), min(a[3], b[3])
    width = max(f(f(right - left) + one), zero)
    height = max(f(f(bottom - top) + one), zero)
    inter = f(width * height)
    sa = f(f(f(a[2] - a[0]) + one) * f(f(a[3] - a[1]) + one))
    sb = f(f(f(b[2] - b[0]) + one) * f(f(b[3] - b[1]) + one))
    return f(inter / f(f(sa + sb) - inter))


def nms_mask(boxes, thresh):
    """nms_kernel, nms_gpu.py:73-108: mask[i, j // 64] bit (j % 64) = IoU(i, j) > thresh for j after i (score-sorted boxes)."""
    n = boxes.shape[0]
    col_blocks = (n + 63) // 64
    mask = np.zeros((n, col_blocks), np.uint64)
    thresh = np.float32(thresh)
    b = boxes.astype(np.float32)
    for i in range(n):
        for j in range(i + 1, n):                                    # start = tx + 1 on the diagonal block (:96-98)
            if nms_iou(b[i], b[j]) > thresh:
                mask[i, j // 64] |= np.uint64(1) << np.uint64(j % 64)
    return mask


def nms_postprocess(keep_out, mask_host, boxes_num):
    """nms_gpu.py:111-127 (the sequential sweep).  Returns the number kept; keep_out[:n] = kept sorted positions."""
    col_blocks = (boxes_num + 63) // 64
    remv = np.zeros(col_blocks, np.uint64)
    num_to_keep = 0
    for i in range(boxes_num):
        nblock, inblock = i // 64, i % 64
        if not (remv[nblock] & (np.uint64(1) << np.uint64(inblock))):
            keep_out[num_to_keep] = i
            num_to_keep += 1
            for j in range(nblock, col_blocks):
                remv[j] |= mask_host[i * col_blocks + j]
    return num_to_keep


def nms_gpu(dets, nms_overlap_thresh):
    """nms_gpu, nms_gpu.py:130-164.  dets [N,5] = (x1,y1,x2,y2,score).  Returns the kept ORIGINAL indices, best score first.
    The source sorts with ``scores.argsort()[::-1]`` (numpy's default sort, tie order unspecified); the restatement and the
    HIP path both use the stable sort reversed (ties: higher index first)."""
    dets = np.asarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = dets[:, 4].argsort(kind="stable")[::-1]
    boxes = dets[order]
    mask = nms_mask(boxes, nms_overlap_thresh)
    keep = np.zeros(n, np.int32)
    num = nms_postprocess(keep, mask.reshape(-1), n)
    return list(order[keep[:num]])


def nms_vectorised(dets, thresh):
    """Same result as nms_gpu with each kept row's IoUs computed by numpy fp32 array ops (for sizes the loops cannot reach)."""
    dets = np.asarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    order = dets[:, 4].argsort(kind="stable")[::-1]
    b = dets[order]
    one = np.float32(1.0)
    area = ((b[:, 2] - b[:, 0]) + one) * ((b[:, 3] - b[:, 1]) + one)
    removed = np.zeros(n, bool)
    keep = []
    thr = np.float32(thresh)
    for i in range(n):                                       # rows of removed boxes are never read by the sweep: skip them
        if removed[i]:
            continue
        keep.append(i)
        r = b[i + 1:]
        w = np.maximum((np.minimum(b[i, 2], r[:, 2]) - np.maximum(b[i, 0], r[:, 0])) + one, np.float32(0))
        h = np.maximum((np.minimum(b[i, 3], r[:, 3]) - np.maximum(b[i, 1], r[:, 1])) + one, np.float32(0))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            removed[i + 1:] |= inter / ((area[i] + area[i + 1:]) - inter) > thr
    return list(order[np.asarray(keep, np.int64)])


# ---------------------------------------------------------------------------------------------------------------------
# rotated-box IoU / NMS, nms_gpu.py:179-653 (numba.cuda in the reference)
# Typing follows the numba source: fp32 for corners / clipping, Python-float (fp64) for the area accumulator and the quotient.
# ---------------------------------------------------------------------------------------------------------------------
_f = np.float32


def rbbox_to_corners(rbbox):
    """:366-389.  rbbox = (x, y, x_d, y_d, angle) -> 8 floats, clockwise corners rotated clockwise."""
    a_cos, a_sin = _f(np.cos(_f(rbbox[4]))), _f(np.sin(_f(rbbox[4])))
    hx, hy = _f(_f(rbbox[2]) / _f(2)), _f(_f(rbbox[3]) / _f(2))
    cx = [-hx, -hx, hx, hx]
    cy = [-hy, hy, hy, -hy]
    c = np.zeros(8, np.float32)
    for i in range(4):
        c[2 * i] = _f(_f(_f(a_cos * cx[i]) + _f(a_sin * cy[i])) + _f(rbbox[0]))
        c[2 * i + 1] = _f(_f(_f(-a_sin * cx[i]) + _f(a_cos * cy[i])) + _f(rbbox[1]))
    return c


def _point_in_quadrilateral(px, py, c, _f=_f):
    """:323-339"""
    ab0, ab1, ad0, ad1 = _f(c[2] - c[0]), _f(c[3] - c[1]), _f(c[6] - c[0]), _f(c[7] - c[1])
    ap0, ap1 = _f(px - c[0]), _f(py - c[1])
    abab = _f(_f(ab0 * ab0) + _f(ab1 * ab1)); abap = _f(_f(ab0 * ap0) + _f(ab1 * ap1))
    adad = _f(_f(ad0 * ad0) + _f(ad1 * ad1)); adap = _f(_f(ad0 * ap0) + _f(ad1 * ap1))
    return abab >= abap and abap >= 0 and adad >= adap and adap >= 0


def _line_segment_intersection(p1, p2, i, j, _f=_f):
    """:235-278.  Returns the intersection point or None."""
    A0, A1, B0, B1 = p1[2 * i], p1[2 * i + 1], p1[2 * ((i + 1) % 4)], p1[2 * ((i + 1) % 4) + 1]
    C0, C1, D0, D1 = p2[2 * j], p2[2 * j + 1], p2[2 * ((j + 1) % 4)], p2[2 * ((j + 1) % 4) + 1]
    BA0, BA1, DA0, CA0, DA1, CA1 = _f(B0 - A0), _f(B1 - A1), _f(D0 - A0), _f(C0 - A0), _f(D1 - A1), _f(C1 - A1)
    acd = _f(DA1 * CA0) > _f(CA1 * DA0)
    bcd = _f(_f(D1 - B1) * _f(C0 - B0)) > _f(_f(C1 - B1) * _f(D0 - B0))
    if acd != bcd:
        abc = _f(CA1 * BA0) > _f(BA1 * CA0)
        abd = _f(DA1 * BA0) > _f(BA1 * DA0)
        if abc != abd:
            DC0, DC1 = _f(D0 - C0), _f(D1 - C1)
            ABBA = _f(_f(A0 * B1) - _f(B0 * A1)); CDDC = _f(_f(C0 * D1) - _f(D0 * C1))
            DH = _f(_f(BA1 * DC0) - _f(BA0 * DC1))
            Dx = _f(_f(ABBA * DC0) - _f(BA0 * CDDC)); Dy = _f(_f(ABBA * DC1) - _f(BA1 * CDDC))
            with np.errstate(divide="ignore", invalid="ignore"):
                return _f(Dx / DH), _f(Dy / DH)
    return None


def rotate_inter(rbbox1, rbbox2):
    """inter(), :392-406 (quadrilateral_intersection :342-363, sort_vertex_in_convex_polygon :195-232, area :185-192).  The source's
    scratch holds 8 points; degenerate overlaps can produce more candidates (it would write past the array) -- here up to 16 are kept."""
    return quad_inter(rbbox_to_corners(rbbox1), rbbox_to_corners(rbbox2), _f)


def quad_inter(c1, c2, _f=_f):
    """The source's intersection-area routine on two corner lists (8 floats each), with every intermediate rounded by ``_f``: np.float32 is
    the numba typing (rotate_inter); np.float64 serves as the geometric value for rbbox_iou (boost::geometry in the reference)."""
    pts = []
    for i in range(4):
        if _point_in_quadrilateral(c1[2 * i], c1[2 * i + 1], c2, _f):
            pts.append([c1[2 * i], c1[2 * i + 1]])
        if _point_in_quadrilateral(c2[2 * i], c2[2 * i + 1], c1, _f):
            pts.append([c2[2 * i], c2[2 * i + 1]])
    for i in range(4):
        for j in range(4):
            t = _line_segment_intersection(c1, c2, i, j, _f)
            if t is not None:
                pts.append([t[0], t[1]])
    pts = pts[:16]
    n = len(pts)
    if n > 0:
        ctr0, ctr1 = _f(0), _f(0)
        for q in pts:
            ctr0 = _f(ctr0 + q[0]); ctr1 = _f(ctr1 + q[1])
        ctr0 = _f(ctr0 / _f(n)); ctr1 = _f(ctr1 / _f(n))
        vs = []
        with np.errstate(divide="ignore", invalid="ignore"):
            for q in pts:
                v0, v1 = _f(q[0] - ctr0), _f(q[1] - ctr1)
                d = _f(np.sqrt(_f(_f(v0 * v0) + _f(v1 * v1))))
                v0, v1 = _f(v0 / d), _f(v1 / d)
                if v1 < 0:
                    v0 = _f(_f(-2) - v0)
                vs.append(v0)
        for i in range(1, n):                       # insertion sort exactly as written (:218-232)
            if vs[i - 1] > vs[i]:
                temp, tq = vs[i], pts[i]
                j = i
                while j > 0 and vs[j - 1] > temp:
                    vs[j] = vs[j - 1]; pts[j] = pts[j - 1]
                    j -= 1
                vs[j] = temp; pts[j] = tq
    area = 0.0
    for i in range(n - 2):
        a, b, c = pts[0], pts[i + 1], pts[i + 2]
        num = _f(_f(_f(a[0] - c[0]) * _f(b[1] - c[1])) - _f(_f(a[1] - c[1]) * _f(b[0] - c[0])))
        area += abs(float(num) / 2.0)
    return area


def rotate_iou_eval(rbox1, rbox2, criterion=-1):
    """devRotateIoU :409-414 / devRotateIoUEval :562-574"""
    area1, area2 = _f(_f(rbox1[2]) * _f(rbox1[3])), _f(_f(rbox2[2]) * _f(rbox2[3]))
    ai = rotate_inter(rbox1, rbox2)
    with np.errstate(divide="ignore", invalid="ignore"):
        if criterion == -1:
            return np.float64(ai) / (np.float64(_f(area1 + area2)) - ai)
        if criterion == 0:
            return np.float64(ai) / np.float64(area1)
        if criterion == 1:
            return np.float64(ai) / np.float64(area2)
    return np.float64(ai)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1):
    """:618-653 (criterion -1 = rotate_iou_gpu :524-559): iou[n, k] = devRotateIoUEval(query[k], boxes[n])."""
    boxes, query_boxes = np.asarray(boxes, np.float32), np.asarray(query_boxes, np.float32)
    out = np.zeros((boxes.shape[0], query_boxes.shape[0]), np.float32)
    for n in range(boxes.shape[0]):
        for k in range(query_boxes.shape[0]):
            out[n, k] = np.float32(rotate_iou_eval(query_boxes[k], boxes[n], criterion))
    return out


def rotate_nms_gpu(dets, nms_overlap_thresh, return_ious=False):
    """:453-488.  dets [N,6] = (x, y, x_d, y_d, angle, score) -> kept original indices, best score first (stable sort reversed)."""
    dets = np.asarray(dets, np.float32)
    n = dets.shape[0]
    if n == 0:
        return ([], {}) if return_ious else []
    order = dets[:, 5].argsort(kind="stable")[::-1]
    b = dets[order]
    col_blocks = (n + 63) // 64
    mask = np.zeros((n, col_blocks), np.uint64)
    thr = np.float64(np.float32(nms_overlap_thresh))
    ious = {}
    for i in range(n):
        for j in range(i + 1, n):
            v = rotate_iou_eval(b[i, :5], b[j, :5], -1)
            ious[(i, j)] = float(v)
            if v > thr:
                mask[i, j // 64] |= np.uint64(1) << np.uint64(j % 64)
    keep = np.zeros(n, np.int32)
    num = nms_postprocess(keep, mask.reshape(-1), n)
    res = list(order[keep[:num]])
    return (res, ious) if return_ious else res


# ---------------------------------------------------------------------------------------------------------------------
# riou_cc / rbbox_iou: libs/ops/box_np_ops.py:16-27 over libs/ops/cc/box_ops.h:23-80.  The C++ there hands the two
# polygons to boost::geometry (intersection, union_, area) -- an un-vendored dependency; what it computes for convex quadrilaterals
# is |P n Q| / |P u Q| with |P u Q| = |P| + |Q| - |P n Q|.  The geometric intersection area is restated as the convex hull (qhull, via
# scipy) of {corners of P in Q} u {corners of Q in P} u {edge crossings} in float64 with a 1e-9 tolerance on the closed tests, so that
# touching and coincident edges -- where the numba routine above returns rounding noise -- have their geometric value, as in boost.
# ---------------------------------------------------------------------------------------------------------------------
def convex_quad_inter_area(p, q, tol=1e-9):
    """area of the intersection of two convex quadrilaterals given as 8 floats (x0, y0, ..., x3, y3), float64"""
    from scipy.spatial import ConvexHull, QhullError
    P = np.asarray(p, np.float64).reshape(4, 2)
    Q = np.asarray(q, np.float64).reshape(4, 2)
    scale = max(1.0, float(np.abs(P).max()), float(np.abs(Q).max()))
    eps = tol * scale * scale

    def inside(pt, poly):
        sgn = []
        for i in range(4):
            a, b = poly[i], poly[(i + 1) % 4]
            sgn.append((b[0] - a[0]) * (pt[1] - a[1]) - (b[1] - a[1]) * (pt[0] - a[0]))
        return all(v >= -eps for v in sgn) or all(v <= eps for v in sgn)

    pts = [c for c in P if inside(c, Q)] + [c for c in Q if inside(c, P)]
    for i in range(4):
        a, b = P[i], P[(i + 1) % 4]
        for j in range(4):
            c, d = Q[j], Q[(j + 1) % 4]
            den = (b[0] - a[0]) * (d[1] - c[1]) - (b[1] - a[1]) * (d[0] - c[0])
            if abs(den) <= eps:
                continue                      # parallel edges: their overlap, if any, ends in corners already listed
            t = ((c[0] - a[0]) * (d[1] - c[1]) - (c[1] - a[1]) * (d[0] - c[0])) / den
            u = ((c[0] - a[0]) * (b[1] - a[1]) - (c[1] - a[1]) * (b[0] - a[0])) / den
            if -tol <= t <= 1 + tol and -tol <= u <= 1 + tol:
                pts.append(a + t * (b - a))
    if len(pts) < 3:
        return 0.0
    try:
        return float(ConvexHull(np.asarray(pts), qhull_options="QJ Pp").volume)
    except QhullError:
        return 0.0


def center_to_corner_box2d(centers, dims, angles, origin=0.5):
    """box_np_ops.py:363-383 with corners_nd :170-201 and rotation_2d :302-315 (clockwise corners from the minimum point, rotated
    clockwise for positive angles), in the dtype of the inputs."""
    dt = dims.dtype
    corners_norm = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dt) - np.array(origin, dt)
    corners = dims.reshape(-1, 1, 2) * corners_norm.reshape(1, 4, 2)
    rot_sin, rot_cos = np.sin(angles), np.cos(angles)
    rot_mat_T = np.stack([[rot_cos, -rot_sin], [rot_sin, rot_cos]])
    corners = np.einsum("aij,jka->aik", corners, rot_mat_T)
    return corners + centers.reshape(-1, 1, 2)


def corner_to_standup_nd(boxes_corner):
    """box_np_ops.py:236-241"""
    return np.concatenate([np.min(boxes_corner, axis=1), np.max(boxes_corner, axis=1)], -1)


def iou_jit(boxes, query_boxes, eps=0.0):
    """box_np_ops.py:654-682"""
    N, K = boxes.shape[0], query_boxes.shape[0]
    overlaps = np.zeros((N, K), boxes.dtype)
    for k in range(K):
        box_area = (query_boxes[k, 2] - query_boxes[k, 0] + eps) * (query_boxes[k, 3] - query_boxes[k, 1] + eps)
        for n in range(N):
            iw = min(boxes[n, 2], query_boxes[k, 2]) - max(boxes[n, 0], query_boxes[k, 0]) + eps
            if iw > 0:
                ih = min(boxes[n, 3], query_boxes[k, 3]) - max(boxes[n, 1], query_boxes[k, 1]) + eps
                if ih > 0:
                    ua = (boxes[n, 2] - boxes[n, 0] + eps) * (boxes[n, 3] - boxes[n, 1] + eps) + box_area - iw * ih
                    overlaps[n, k] = iw * ih / ua
    return overlaps


def _shoelace(c):
    x, y = c[0::2], c[1::2]
    return 0.5 * abs(sum(float(x[i]) * float(y[(i + 1) % 4]) - float(x[(i + 1) % 4]) * float(y[i]) for i in range(4)))


def rbbox_iou(box_corners, qbox_corners, standup_iou, standup_thresh):
    """cc/box_ops.h:23-80: overlaps[n, k] for the pairs with standup_iou[n, k] > standup_thresh, 0 elsewhere."""
    N, K = box_corners.shape[0], qbox_corners.shape[0]
    out = np.zeros((N, K), box_corners.dtype)
    for k in range(K):
        for n in range(N):
            if standup_iou[n, k] <= standup_thresh:
                continue
            p = np.asarray(box_corners[n], np.float64).reshape(8)
            q = np.asarray(qbox_corners[k], np.float64).reshape(8)
            ai = convex_quad_inter_area(p, q)
            if ai > 0:
                un = _shoelace(p) + _shoelace(q) - ai
                if un > 0:
                    out[n, k] = ai / un
    return out


def riou_cc(rbboxes, qrbboxes, standup_thresh=0.0):
    """box_np_ops.py:16-27"""
    bc = center_to_corner_box2d(rbboxes[:, :2], rbboxes[:, 2:4], rbboxes[:, 4])
    qc = center_to_corner_box2d(qrbboxes[:, :2], qrbboxes[:, 2:4], qrbboxes[:, 4])
    su = iou_jit(corner_to_standup_nd(bc), corner_to_standup_nd(qc), eps=0.0)
    return rbbox_iou(bc, qc, su, standup_thresh)
