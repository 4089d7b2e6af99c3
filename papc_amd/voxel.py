"""The steps either side of the PillarFeatureNet, with the reference's signatures.

  points_to_voxel      <- /root/reference/PAPC/models/detect/pointpillars/libs/ops/point_cloud/point_cloud_ops.py:106-166
  PointPillarsScatter  <- /root/reference/PAPC/models/detect/pointpillars/models/bones/pillars.py:110-142

Both run in libpapc_hip.so (csrc/voxel.hip); there is no CPU fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000, padded=False):
    """points [N, ndim>=3] float32 CUDA tensor -> (voxels [M, max_points, ndim], coordinates [M, 3] int32 (zyx when
    ``reverse_index``), num_points_per_voxel [M] int32), M = number of voxels created (<= max_voxels).

    Exactly the reference's sequential first-come assignment (voxel numbers in order of first appearance, points in
    input order, ``break`` once ``max_voxels`` voxels exist).  Slicing to M needs one device->host read of the voxel
    count; ``padded=True`` returns the full ``max_voxels``-row buffers plus the device-side count instead (no sync):
    (voxels, coors, num_points, voxel_num [1] int32)."""
    if not (isinstance(points, torch.Tensor) and points.is_cuda):
        raise _lib.PapcError("points_to_voxel needs a CUDA (ROCm) tensor: there is no CPU fallback")
    pts = points.contiguous().float()
    N, ndim = pts.shape
    dev = pts.device
    vs = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    cr = (ctypes.c_float * 6)(*[float(v) for v in coors_range])
    voxels = torch.empty(max_voxels, max_points, ndim, device=dev, dtype=torch.float32)
    coors = torch.empty(max_voxels, 3, device=dev, dtype=torch.int32)
    nump = torch.empty(max_voxels, device=dev, dtype=torch.int32)
    vnum = torch.empty(1, device=dev, dtype=torch.int32)
    lib = _lib.load()
    if N == 0:
        voxels.zero_(); coors.zero_(); nump.zero_(); vnum.zero_()
    else:
        wbytes = lib.papc_points_to_voxel_workspace(N)
        work = torch.empty(wbytes, device=dev, dtype=torch.uint8)
        check(lib.papc_points_to_voxel_f32(ptr(pts), N, ndim, ctypes.cast(vs, ctypes.c_void_p), ctypes.cast(cr, ctypes.c_void_p),
                                           int(max_points), int(max_voxels), int(bool(reverse_index)), ptr(voxels), ptr(coors),
                                           ptr(nump), ptr(vnum), ptr(work), wbytes, stream_ptr()), "papc_points_to_voxel_f32")
    if padded:
        return voxels, coors, nump, vnum
    m = int(vnum.item())
    return voxels[:m], coors[:m], nump[:m]


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, coords, batch_size, ny, nx):
        feats = feats.contiguous().float()
        coords = coords.to(torch.int32).contiguous()
        P, C = feats.shape
        dev = feats.device
        canvas = torch.empty(batch_size, C, ny, nx, device=dev, dtype=torch.float32)
        owner = torch.empty(batch_size, ny, nx, device=dev, dtype=torch.int32)
        check(_lib.load().papc_pillar_scatter_f32(ptr(feats), ptr(coords), P, C, batch_size, ny, nx, ptr(canvas), ptr(owner),
                                                  stream_ptr()), "papc_pillar_scatter_f32")
        ctx.save_for_backward(coords, owner)
        ctx.dims = (P, C, batch_size, ny, nx)
        return canvas

    @staticmethod
    def backward(ctx, g):
        coords, owner = ctx.saved_tensors
        P, C, B, ny, nx = ctx.dims
        g = g.contiguous().float()
        gf = torch.empty(P, C, device=g.device, dtype=torch.float32)
        if P > 0:
            check(_lib.load().papc_pillar_scatter_bwd_f32(ptr(g), ptr(coords), ptr(owner), P, C, B, ny, nx, ptr(gf), stream_ptr()),
                  "papc_pillar_scatter_bwd_f32")
        return gf, None, None, None, None


class PointPillarsScatter(nn.Module):
    """pillars.py:110-142.  forward(voxel_features [P, C], coords [P, 4] = (batch, z, y, x), batch_size) ->
    [batch_size, C, ny, nx] (``output_shape`` = [_, _, ny, nx] like the source)."""

    def __init__(self, output_shape, num_input_features=4):
        super().__init__()
        self.name = 'PointPillarsScatter'
        self.output_shape = output_shape
        self.ny = int(output_shape[2])
        self.nx = int(output_shape[3])
        self.nchannels = num_input_features

    def forward(self, voxel_features, coords, batch_size):
        if not voxel_features.is_cuda:
            raise _lib.PapcError("PointPillarsScatter needs CUDA (ROCm) tensors: there is no CPU fallback")
        assert voxel_features.shape[1] == self.nchannels
        return _Scatter.apply(voxel_features, coords, int(batch_size), self.ny, self.nx)
