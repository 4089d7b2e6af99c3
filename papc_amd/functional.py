"""The reference's free functions, same names / argument order / layouts, on PyTorch-ROCm tensors.

Mirrors /root/reference/PAPC/models/layers/pointnet2_basic_layers.py (``square_distance`` :26,
``index_points`` :43, ``farthest_point_sample`` :65, ``query_ball_point`` :98, ``sample_and_group`` :129,
``sample_and_group_all`` :160).  Every function launches hand-written gfx950 kernels through the C ABI in
``include/papc_hip.h``; there is no eager/CPU fallback -- CPU tensors raise.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PapcError("papc_amd ops need CUDA(ROCm) tensors; got a %s tensor (no CPU fallback)" % t.device)


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def radius_threshold(radius):
    """``radius ** 2`` as the reference's comparison sees it: a python double rounded to fp32 (:112)."""
    return ctypes.c_float(float(radius) * float(radius)).value


def square_distance(src, dst):
    """:26-40.  src [B,N,3], dst [B,M,3] -> [B,N,M] (API parity; the hot path never builds this matrix)."""
    _need_cuda(src, dst)
    src, dst = _f32c(src), _f32c(dst)
    B, N, C = src.shape
    M = dst.shape[1]
    if C != 3 or dst.shape[2] != 3:
        raise _lib.PapcError("square_distance: only C=3 is built (got %d)" % C)
    out = torch.empty(B, N, M, device=src.device, dtype=torch.float32)
    check(_lib.load().papc_square_distance_f32(ptr(src), ptr(dst), B, N, M, ptr(out), stream_ptr()), "papc_square_distance_f32")
    return out


def index_points(points, idx):
    """:43-62.  points [B,N,C], idx [B,S] or [B,S,K] (any integer or float dtype, like the source) ->
    [B,S,C] / [B,S,K,C].  Differentiable w.r.t. ``points`` (the reference cuts autograd here, :57-60)."""
    return _IndexPoints.apply(points, idx)


class _IndexPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        _need_cuda(points, idx)
        points = _f32c(points)
        if idx.dtype not in (torch.int32, torch.int64):
            idx = idx.to(torch.int64)  # idx_np.astype('int64')  (:59)
        idx = idx.contiguous()
        B, N, C = points.shape
        S = idx[0].numel()
        out = torch.empty(tuple(idx.shape) + (C,), device=points.device, dtype=torch.float32)
        check(_lib.load().papc_index_points_f32(ptr(points), ptr(idx), int(idx.dtype == torch.int64), B, N, C, S,
                                                ptr(out), stream_ptr()), "papc_index_points_f32")
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, C, S)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, C, S = ctx.shape
        g = _f32c(g)
        gp = torch.zeros(B, N, C, device=g.device, dtype=torch.float32)
        check(_lib.load().papc_index_points_bwd_f32(ptr(g), ptr(idx), int(idx.dtype == torch.int64), B, N, C, S,
                                                    ptr(gp), stream_ptr()), "papc_index_points_bwd_f32")
        return gp, None


def _fps_raw(xyz, npoint, start_idx, init_dist=1.0, want_new_xyz=True, new_xyz_out=None):
    """xyz [B,N,3] (any strides, e.g. a transposed [B,3,N]) -> (idx int32 [B,npoint], new_xyz [B,npoint,3]).
    ``new_xyz_out``: optional preallocated contiguous float32 [B,npoint,3] the centroids are written to (a static buffer of a
    captured training step)."""
    _need_cuda(xyz)
    if xyz.dtype != torch.float32:
        xyz = xyz.float()
    B, N, C = xyz.shape
    assert C == 3
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), device=xyz.device, dtype=torch.int64)  # paddle.randint (:76)
    start_idx = start_idx.to(device=xyz.device, dtype=torch.int64).contiguous()
    idx = torch.empty(B, npoint, device=xyz.device, dtype=torch.int32)
    new_xyz = torch.empty(B, npoint, 3, device=xyz.device, dtype=torch.float32) if want_new_xyz else None
    if new_xyz_out is not None:
        assert tuple(new_xyz_out.shape) == (B, npoint, 3) and new_xyz_out.is_contiguous() and new_xyz_out.dtype == torch.float32
        new_xyz = new_xyz_out
    check(_lib.load().papc_fps_f32(ptr(xyz), xyz.stride(0), xyz.stride(1), xyz.stride(2), B, N, npoint,
                                   ptr(start_idx), float(init_dist), ptr(idx), ptr(new_xyz), stream_ptr()), "papc_fps_f32")
    return idx, new_xyz


def farthest_point_sample(xyz, npoint, start_idx=None, init_dist=1.0, as_float=False):
    """:65-95.  xyz [B,N,3] -> sampled indices [B,npoint].

    ``start_idx`` [B] pins the first centroid (the source draws it with paddle.randint, :76; ``None`` draws
    it with torch.randint).  ``init_dist`` is the source's running-distance init (1.0, :75).  The source
    returns float32 (``paddle.zeros`` :74): ``as_float=True`` reproduces that dtype; the default is int64.
    """
    idx, _ = _fps_raw(xyz, npoint, start_idx, init_dist, want_new_xyz=False)
    return idx.float() if as_float else idx.long()


def _ball_query_raw(radii, nsamples, xyz, new_xyz, idx64=False, outs=None):
    """All radii in one scan.  xyz [B,N,3] strided, new_xyz [B,S,3] -> list of [B,S,K_r] (int32 / int64).
    ``outs``: optional preallocated contiguous index tensors, one per radius."""
    _need_cuda(xyz, new_xyz)
    if xyz.dtype != torch.float32:
        xyz = xyz.float()
    new_xyz = _f32c(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    n = len(radii)
    dt = torch.int64 if idx64 else torch.int32
    if outs is None:
        outs = [torch.empty(B, S, int(k), device=xyz.device, dtype=dt) for k in nsamples]
    else:
        assert len(outs) == n and all(tuple(o.shape) == (B, S, int(k)) and o.dtype == dt and o.is_contiguous() for o, k in zip(outs, nsamples))
    thr = (ctypes.c_float * n)(*[radius_threshold(r) for r in radii])
    ns = (ctypes.c_int * n)(*[int(k) for k in nsamples])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    check(_lib.load().papc_ball_query_f32(ptr(xyz), xyz.stride(0), xyz.stride(1), xyz.stride(2), ptr(new_xyz), B, N, S,
                                          n, ctypes.cast(thr, ctypes.c_void_p), ctypes.cast(ns, ctypes.c_void_p),
                                          ctypes.cast(op, ctypes.c_void_p), int(idx64), stream_ptr()), "papc_ball_query_f32")
    return outs


def query_ball_point(radius, nsample, xyz, new_xyz):
    """:98-126.  -> group_idx [B,S,nsample] int64 (the source's dtype)."""
    return _ball_query_raw([radius], [nsample], xyz, new_xyz, idx64=True)[0]


def _group_raw(xyz, new_xyz, feats, idx, xyz_first=True):
    """[B,S,K,3+D] = concat(xyz[idx]-new_xyz, feats[idx]) (or feats first).  idx int32 [B,S,K]."""
    B, N, _ = xyz.shape
    S, K = idx.shape[1], idx.shape[2]
    D = 0 if feats is None else feats.shape[2]
    if feats is not None:
        feats = _f32c(feats)
    out = torch.empty(B, S, K, 3 + D, device=xyz.device, dtype=torch.float32)
    check(_lib.load().papc_group_points_f32(ptr(xyz), xyz.stride(0), xyz.stride(1), xyz.stride(2), ptr(new_xyz), ptr(feats),
                                            ptr(idx), B, N, S, K, D, int(xyz_first), ptr(out), stream_ptr()), "papc_group_points_f32")
    return out


class _GroupPoints(torch.autograd.Function):
    """_group_raw with the gradient let through (the reference's index_points cuts it, :57-60): to the gathered features, the gathered
    coordinates and the centres."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats, idx, xyz_first):
        ctx.save_for_backward(idx)
        ctx.shape = (xyz.shape[0], xyz.shape[1], 0 if feats is None else feats.shape[2], bool(xyz_first))
        return _group_raw(xyz, new_xyz, feats, idx, xyz_first)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, D, xyz_first = ctx.shape
        S, K = idx.shape[1], idx.shape[2]
        g = _f32c(g)
        need_x, need_c, need_f = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2] and D > 0
        gx = torch.zeros(B, N, 3, device=g.device, dtype=torch.float32) if need_x else None
        gc = torch.empty(B, S, 3, device=g.device, dtype=torch.float32) if need_c else None
        gf = torch.zeros(B, N, D, device=g.device, dtype=torch.float32) if need_f else None
        check(_lib.load().papc_group_points_bwd_f32(ptr(g), ptr(idx), B, N, S, K, D, int(xyz_first), ptr(gf), ptr(gx), ptr(gc), stream_ptr()),
              "papc_group_points_bwd_f32")
        return gx, gc, gf, None, None


def group_points(xyz, new_xyz, feats, idx, xyz_first=True):
    """[B,S,K,3+D] = concat(xyz[idx] - new_xyz, feats[idx]) (SSG order, :146-153; ``xyz_first=False``: the MSG order :263-269), idx int32
    [B,S,K].  Differentiable where an input asks for it; without that it is the plain gather."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (xyz, new_xyz, feats)):
        return _GroupPoints.apply(xyz, new_xyz, feats, idx, xyz_first)
    return _group_raw(xyz, new_xyz, feats, idx, xyz_first)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, start_idx=None, init_dist=1.0):
    """:129-157.  xyz [B,N,3], points [B,N,D] or None -> new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D]
    (xyz-normalised coordinates first, :151)."""
    _need_cuda(xyz, points)
    if xyz.dtype != torch.float32:
        xyz = xyz.float()
    fps_idx, new_xyz = _fps_raw(xyz, npoint, start_idx, init_dist)                 # :143-144
    idx = _ball_query_raw([radius], [nsample], xyz, new_xyz)[0]                    # :145
    new_points = group_points(xyz, new_xyz, points, idx, xyz_first=True)           # :146-153
    if returnfps:
        grouped_xyz = index_points(xyz.contiguous(), idx)
        return new_xyz, new_points, grouped_xyz, fps_idx.long()
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """:160-176 (pure reshape/concat; raw xyz, not centred)."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device, dtype=xyz.dtype)
    grouped_xyz = xyz.reshape(B, 1, N, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz, points.reshape(B, 1, N, -1)], dim=-1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points


def three_nn(xyz1, xyz2, out=None):
    """The neighbour search of PointNetFeaturePropagation.forward (:315-322) in one kernel: xyz1 [B,N,3], xyz2 [B,S,3]
    (any strides, S >= 3) -> (dist3 [B,N,3] the three smallest ``square_distance`` values ascending, idx3 [B,N,3]
    int32 the TRUE neighbour indices (stable order), weight3 [B,N,3] = (1/(d+1e-8)) / sum).  No [B,N,S] matrix and
    no sort is materialised."""
    _need_cuda(xyz1, xyz2)
    xyz1 = xyz1 if xyz1.dtype == torch.float32 else xyz1.float()
    xyz2 = xyz2 if xyz2.dtype == torch.float32 else xyz2.float()
    B, N, C = xyz1.shape
    S = xyz2.shape[1]
    assert C == 3 and xyz2.shape[0] == B and xyz2.shape[2] == 3
    if out is not None:       # preallocated (dist3, idx3, w3): a captured step's plan buffers
        dist3, idx3, w3 = out
    else:
        dist3 = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.float32)
        idx3 = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.int32)
        w3 = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.float32)
    check(_lib.load().papc_three_nn_f32(ptr(xyz1), xyz1.stride(0), xyz1.stride(1), xyz1.stride(2), ptr(xyz2), xyz2.stride(0),
                                        xyz2.stride(1), xyz2.stride(2), B, N, S, ptr(dist3), ptr(idx3), ptr(w3), stream_ptr()),
          "papc_three_nn_f32")
    return dist3, idx3, w3


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points2, idx3, weight3, first3=False):
        _need_cuda(points2, idx3, weight3)
        points2 = _f32c(points2)
        if idx3.dtype != torch.int32 or not idx3.is_contiguous():
            idx3 = idx3.to(torch.int32).contiguous()
        weight3 = _f32c(weight3)
        B, S, D = points2.shape
        N = idx3.shape[1]
        out = torch.empty(B, N, D, device=points2.device, dtype=torch.float32)
        check(_lib.load().papc_three_interpolate_f32(ptr(points2), ptr(idx3), ptr(weight3), B, N, S, D, ptr(out), stream_ptr()),
              "papc_three_interpolate_f32")
        ctx.save_for_backward(idx3, weight3)
        ctx.shape = (B, N, S, D)
        ctx.first3 = bool(first3) and S >= 3
        return out

    @staticmethod
    def backward(ctx, g):
        idx3, weight3 = ctx.saved_tensors
        B, N, S, D = ctx.shape
        g = _f32c(g)
        if ctx.first3:       # every query's neighbours are support points 0, 1, 2: a column reduction per cloud, no atomics, no fill
            gp = torch.empty(B, S, D, device=g.device, dtype=torch.float32)
            check(_lib.load().papc_three_interpolate_bwd_first3_f32(ptr(g), ptr(weight3), B, N, S, D, ptr(gp), stream_ptr()),
                  "papc_three_interpolate_bwd_first3_f32")
            return gp, None, None, None
        gp = _lib.zeros((B, S, D), g.device)
        check(_lib.load().papc_three_interpolate_bwd_f32(ptr(g), ptr(idx3), ptr(weight3), B, N, S, D, ptr(gp), stream_ptr()),
              "papc_three_interpolate_bwd_f32")
        return gp, None, None, None


def three_interpolate(points2, idx3, weight3, first3=False):
    """sum(index_points(points2, idx) * weight[..., None], axis=2)  (:323): points2 [B,S,D], idx3/weight3 [B,N,3] ->
    [B,N,D].  Differentiable w.r.t. ``points2`` (the weights depend on coordinates only).  ``first3=True``: the caller vouches that idx3 is
    the constant (0, 1, 2) of every query (the reference's sort-then-argsort neighbours): the backward is then a reduction, not a scatter."""
    return _ThreeInterpolate.apply(points2, idx3, weight3, first3)
