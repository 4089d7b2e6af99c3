"""PointPillars' PillarFeatureNet with the reference's constructor / forward signatures.

Mirrors /root/reference/PAPC/models/detect/pointpillars/models/bones/pillars.py: ``PFNLayer`` :9-41 and
``PillarFeatureNet`` :43-108 (caller: models/detectors/pointpillars.py:137 ``self.pfn(voxels, num_points, coors)``).

The shipped configuration (``num_filters: [64]``, pointpillars_kitti_car_xy16.yaml:56) is a single last PFNLayer:
that path is two fused HIP passes (papc_pfn_stats_f32 / papc_pfn_apply_f32) plus their backward.  Longer
``num_filters`` chains run layer by layer on the generic MFMA row-GEMM kernels (forward only for non-last
layers).  ``use_norm=False`` is not built.

paddle's ``nn.Linear.weight`` is ``[in,out]``; here the weight is torch-style ``[out,in]`` (transpose when
importing a paddle checkpoint).  BatchNorm1D(momentum=0.01) in paddle weighs the RUNNING value by 0.01.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr
from .mlp import StackSpec, shared_mlp_max


class _PFNFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, features, num_voxels, coors, w, gamma, beta, rmean, rvar, eps, momentum):
        lib = _lib.load()
        st = stream_ptr()
        P, T, _ = features.shape
        C = w.shape[0]
        vx, vy, xo, yo = geom
        dev = features.device
        nb = lib.papc_pfn_num_blocks(P)
        stats = torch.empty(nb, 2, C, device=dev, dtype=torch.float32)
        check(lib.papc_pfn_stats_f32(ptr(features), ptr(num_voxels), ptr(coors), P, T, vx, vy, xo, yo, ptr(w), C, ptr(stats),
                                     None, st), "papc_pfn_stats_f32")
        cst = torch.empty(4, C, device=dev, dtype=torch.float32)
        check(lib.papc_bn_finalize_f32(ptr(stats), nb, P * T, C, ptr(gamma), ptr(beta), eps, momentum, cst[0].data_ptr(),
                                       cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(), ptr(rmean), ptr(rvar), st),
              "papc_bn_finalize_f32")
        out = torch.empty(P, C, device=dev, dtype=torch.float32)
        argmax = torch.empty(P, C, device=dev, dtype=torch.int32)
        check(lib.papc_pfn_apply_f32(ptr(features), ptr(num_voxels), ptr(coors), P, T, vx, vy, xo, yo, ptr(w), C,
                                     cst[2].data_ptr(), cst[3].data_ptr(), ptr(out), ptr(argmax), st), "papc_pfn_apply_f32")
        ctx.geom = geom
        ctx.save_for_backward(features, num_voxels, coors, w, cst, argmax)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        st = stream_ptr()
        features, num_voxels, coors, w, cst, argmax = ctx.saved_tensors
        vx, vy, xo, yo = ctx.geom
        P, T, _ = features.shape
        C = w.shape[0]
        dev = features.device
        gout = gout.contiguous().float()
        nb = lib.papc_pfn_num_blocks(P)
        red = torch.empty(nb, 2, C, device=dev, dtype=torch.float32)
        geo = (ptr(features), ptr(num_voxels), ptr(coors), P, T, vx, vy, xo, yo, ptr(w), C)
        bn = (cst[0].data_ptr(), cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr())
        check(lib.papc_pfn_bwd_reduce_f32(*geo, ptr(gout), ptr(argmax), *bn, ptr(red), st), "papc_pfn_bwd_reduce_f32")
        dgb = torch.empty(2, C, device=dev, dtype=torch.float32)
        c12 = torch.empty(2, C, device=dev, dtype=torch.float32)
        check(lib.papc_bn_bwd_finalize_f32(ptr(red), nb, P * T, C, dgb[0].data_ptr(), dgb[1].data_ptr(), c12[0].data_ptr(),
                                           c12[1].data_ptr(), 0, st), "papc_bn_bwd_finalize_f32")
        dwp = torch.empty(nb, C, 9, device=dev, dtype=torch.float32)
        check(lib.papc_pfn_bwd_dw_f32(*geo, ptr(gout), ptr(argmax), *bn, c12[0].data_ptr(), c12[1].data_ptr(), ptr(dwp), st),
              "papc_pfn_bwd_dw_f32")
        dw = torch.empty(C, 9, device=dev, dtype=torch.float32)
        check(lib.papc_reduce_partials_f32(ptr(dwp), nb, C * 9, ptr(dw), 0, st), "papc_reduce_partials_f32")
        return None, None, None, None, dw, dgb[0], dgb[1], None, None, None, None


class PFNLayer(nn.Module):
    """pillars.py:9-41."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.name = 'PFNLayer'
        self.last_vfe = last_layer
        if not self.last_vfe:
            out_channels = out_channels // 2                                                   # :18-19
        self.units = out_channels
        if not use_norm:
            raise NotImplementedError("PFNLayer(use_norm=False) is not built (the shipped config uses use_norm=True)")
        self.linear = nn.Linear(in_channels, out_channels, bias=False)                         # :23
        self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)                      # :24 (paddle momentum)

    def forward(self, inputs):
        """inputs [P,T,Cin] -> [P,1,C] (last layer, :36-37) or [P,T,2C] (concat with the tiled max, :39-41)."""
        P, T, Cin = inputs.shape
        C = self.units
        rows = inputs.reshape(P * T, Cin).contiguous().float()
        if self.last_vfe:
            spec = StackSpec(P, T, 1, T, 0, True, eps=self.norm.eps, momentum=self.norm.momentum)
            zero_b = torch.zeros(C, device=rows.device, dtype=torch.float32)
            out = shared_mlp_max(spec, [(self.norm.running_mean, self.norm.running_var)], None, None, None, None,
                                 [self.linear.weight, zero_b, self.norm.weight, self.norm.bias], x_rows=rows)
            return out.view(P, 1, C)
        # non-last layer: activations are an output, so they are materialised (forward only)
        lib = _lib.load()
        st = stream_ptr()
        dev = rows.device
        M = P * T
        with torch.no_grad():
            parts = lib.papc_mlp_gemm_parts(M)
            y = torch.empty(M, C, device=dev, dtype=torch.float32)
            stats = torch.empty(parts, 2, C, device=dev, dtype=torch.float32)
            check(lib.papc_mlp_gemm_f32(0, ptr(rows), Cin, None, None, None, ptr(self.linear.weight), None, M, Cin, C, ptr(y),
                                        ptr(stats), None, st), "papc_mlp_gemm_f32")
            cst = torch.empty(4, C, device=dev, dtype=torch.float32)
            check(lib.papc_bn_finalize_f32(ptr(stats), parts, M, C, ptr(self.norm.weight), ptr(self.norm.bias), self.norm.eps,
                                           self.norm.momentum, cst[0].data_ptr(), cst[1].data_ptr(), cst[2].data_ptr(),
                                           cst[3].data_ptr(), ptr(self.norm.running_mean), ptr(self.norm.running_var), st),
                  "papc_bn_finalize_f32")
            x = torch.empty(M, C, device=dev, dtype=torch.float32)
            check(lib.papc_bn_relu_f32(ptr(y), cst[2].data_ptr(), cst[3].data_ptr(), M, C, ptr(x), st), "papc_bn_relu_f32")
            x_max = torch.empty(P, C, device=dev, dtype=torch.float32)
            check(lib.papc_bn_relu_max_f32(ptr(y), cst[2].data_ptr(), cst[3].data_ptr(), P, T, C, ptr(x_max), None, st),
                  "papc_bn_relu_max_f32")
            x = x.view(P, T, C)
            return torch.cat([x, x_max.view(P, 1, C).expand(P, T, C)], dim=2)                  # :39-41


class PillarFeatureNet(nn.Module):
    """pillars.py:43-108."""

    def __init__(self, num_input_features=4, use_norm=True, num_filters=(64, 128), with_distance=False,
                 voxel_size=(0.2, 0.2, 4), pc_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.name = 'PillarFeatureNet'
        assert len(num_filters) > 0
        if num_input_features != 4 or with_distance:
            raise NotImplementedError("PillarFeatureNet is built for num_input_features=4, with_distance=False "
                                      "(the shipped KITTI config)")
        num_input_features += 5                                                                # :55
        self._with_distance = with_distance
        num_filters = [num_input_features] + list(num_filters)
        layers = []
        for i in range(len(num_filters) - 1):
            layers.append(PFNLayer(num_filters[i], num_filters[i + 1], use_norm, last_layer=(i >= len(num_filters) - 2)))
        self.pfn_layers = nn.ModuleList(layers)                                                # :71
        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]                                              # :76
        self.y_offset = self.vy / 2 + pc_range[1]                                              # :77

    def _geom(self):
        return (float(self.vx), float(self.vy), float(self.x_offset), float(self.y_offset))

    def decorate(self, features, num_voxels, coors):
        """:82-102 only -> masked 9-channel rows [P,T,9]."""
        lib = _lib.load()
        features = features.contiguous().float()
        P, T, _ = features.shape
        out = torch.empty(P, T, 9, device=features.device, dtype=torch.float32)
        vx, vy, xo, yo = self._geom()
        check(lib.papc_pfn_decorate_f32(ptr(features), ptr(num_voxels.int().contiguous()), ptr(coors.int().contiguous()), P, T,
                                        vx, vy, xo, yo, ptr(out), stream_ptr()), "papc_pfn_decorate_f32")
        return out

    def forward(self, features, num_voxels, coors):
        """features [P,T,4] f32, num_voxels [P] int, coors [P,4] int (batch,z,y,x) -> [P,C]."""
        if not features.is_cuda:
            raise _lib.PapcError("PillarFeatureNet needs CUDA(ROCm) tensors (no CPU fallback)")
        features = features.contiguous().float()
        num_voxels = num_voxels.int().contiguous()
        coors = coors.int().contiguous()
        if len(self.pfn_layers) == 1:
            pfn = self.pfn_layers[0]
            out = _PFNFused.apply(self._geom(), features, num_voxels, coors, pfn.linear.weight, pfn.norm.weight, pfn.norm.bias,
                                  pfn.norm.running_mean, pfn.norm.running_var, pfn.norm.eps, pfn.norm.momentum)
            return out.squeeze()                                                               # :108
        x = self.decorate(features, num_voxels, coors)
        for pfn in self.pfn_layers:                                                            # :105-106
            x = pfn(x)
        return x.squeeze()
