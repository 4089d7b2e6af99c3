"""PointPillars' PillarFeatureNet with the reference's constructor / forward signatures.

Mirrors /root/reference/PAPC/models/detect/pointpillars/models/bones/pillars.py: ``PFNLayer`` :9-41 and
``PillarFeatureNet`` :43-108 (caller: models/detectors/pointpillars.py:137 ``self.pfn(voxels, num_points, coors)``).

The shipped configuration (``num_filters: [64]``, pointpillars_kitti_car_xy16.yaml:56) is a single last PFNLayer:
that path is two fused HIP passes (papc_pfn_stats_f32 / papc_pfn_apply_f32) plus their backward.  Everything else the
constructors allow -- the source's DEFAULT ``num_filters=(64, 128)`` chain, ``use_norm=False`` (Linear with bias, no norm,
:25-27), ``with_distance=True`` (:57-58, :92-94) -- runs layer by layer on the generic row kernels, differentiable end to
end: the decoration kernel, then per layer the MFMA row GEMM + BN/ReLU stack node (``pool=False``), the group max with its
backward (papc_group_max_bwd_f32) and the concat of :39-41.  ``model.eval()`` switches the norms to their running statistics
(they are registered layers in the source).

paddle's ``nn.Linear.weight`` is ``[in,out]``; here the weight is torch-style ``[out,in]`` (transpose when
importing a paddle checkpoint).  BatchNorm1D(momentum=0.01) in paddle weighs the RUNNING value by 0.01.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr
from .mlp import StackSpec, shared_mlp_max


class PfnDesc(ctypes.Structure):
    """papc_pfn_desc"""
    _fields_ = [("P", ctypes.c_int32), ("T", ctypes.c_int32), ("C", ctypes.c_int32), ("vx", ctypes.c_float), ("vy", ctypes.c_float),
                ("x_offset", ctypes.c_float), ("y_offset", ctypes.c_float), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
                ("training", ctypes.c_int32), ("zero_padded", ctypes.c_int32)]


class PfnIo(ctypes.Structure):
    """papc_pfn_io"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("features", "num_voxels", "coors", "w", "gamma", "beta", "running_mean", "running_var", "out",
                                               "saved", "scratch", "tickets")]


class _PFNFused(torch.autograd.Function):
    """the single-layer PillarFeatureNet (pillars.py:79-108) through the library's coarse entry points: papc_pfn_fwd / papc_pfn_bwd"""

    @staticmethod
    def forward(ctx, geom, features, num_voxels, coors, w, gamma, beta, rmean, rvar, eps, momentum, training=True, tickets=None, zero_padded=False):
        lib = _lib.load()
        P, T, _ = features.shape
        C = w.shape[0]
        dev = features.device
        d = PfnDesc(P, T, C, geom[0], geom[1], geom[2], geom[3], eps, momentum, int(bool(training)), int(bool(zero_padded)))
        sb, wb = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib.papc_pfn_workspace(ctypes.byref(d), ctypes.byref(sb), ctypes.byref(wb)), "papc_pfn_workspace")
        saved = torch.empty(sb.value, device=dev, dtype=torch.uint8)
        scratch = torch.empty(wb.value, device=dev, dtype=torch.uint8)
        out = torch.empty(P, C, device=dev, dtype=torch.float32)
        io = PfnIo(ptr(features), ptr(num_voxels), ptr(coors), ptr(w), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ptr(out), ptr(saved), ptr(scratch),
                   ptr(tickets))
        check(lib.papc_pfn_fwd(ctypes.byref(d), ctypes.byref(io), stream_ptr()), "papc_pfn_fwd")
        ctx.desc, ctx.saved = d, saved
        ctx.bufs = (rmean, rvar, tickets)
        from .mlp import grad_targets_of
        ctx.grad_targets = grad_targets_of([w, gamma, beta]) if torch.is_grad_enabled() or w.requires_grad else None
        ctx.save_for_backward(features, num_voxels, coors, w, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        features, num_voxels, coors, w, gamma, beta = ctx.saved_tensors
        d = ctx.desc
        dev = features.device
        gout = gout.contiguous().float()
        sb, wb = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib.papc_pfn_workspace(ctypes.byref(d), ctypes.byref(sb), ctypes.byref(wb)), "papc_pfn_workspace")
        scratch = torch.empty(wb.value, device=dev, dtype=torch.uint8)
        io = PfnIo(ptr(features), ptr(num_voxels), ptr(coors), ptr(w), ptr(gamma), ptr(beta), ptr(ctx.bufs[0]), ptr(ctx.bufs[1]), None, ptr(ctx.saved),
                   ptr(scratch), ptr(ctx.bufs[2]))
        tg = ctx.grad_targets            # (w.grad, gamma.grad, beta.grad) of parameters that opted in to in-place accumulation, or None
        if tg is not None and any(t is None for t in tg):
            tg = None
        if tg is not None:
            check(lib.papc_pfn_bwd(ctypes.byref(d), ctypes.byref(io), ptr(gout), tg[0].data_ptr(), tg[1].data_ptr(), tg[2].data_ptr(), 1, stream_ptr()),
                  "papc_pfn_bwd")
            return (None,) * 14
        dgb = torch.empty(2, d.C, device=dev, dtype=torch.float32)
        dw = torch.empty(d.C, 9, device=dev, dtype=torch.float32)
        check(lib.papc_pfn_bwd(ctypes.byref(d), ctypes.byref(io), ptr(gout), ptr(dw), dgb[0].data_ptr(), dgb[1].data_ptr(), 0, stream_ptr()), "papc_pfn_bwd")
        return None, None, None, None, dw, dgb[0], dgb[1], None, None, None, None, None, None, None


class _GroupMax(torch.autograd.Function):
    """max over the K rows of each group of NON-NEGATIVE activations (x = relu(.)): [G*K, C] -> [G, C] (pillars.py:34).
    Forward: papc_bn_relu_max_f32 with unit constants (relu is the identity on x >= 0); backward: papc_group_max_bwd_f32."""

    @staticmethod
    def forward(ctx, x, G, K):
        lib = _lib.load()
        C = x.shape[1]
        dev = x.device
        one = torch.ones(C, device=dev, dtype=torch.float32)
        zero = torch.zeros(C, device=dev, dtype=torch.float32)
        out = torch.empty(G, C, device=dev, dtype=torch.float32)
        argmax = torch.empty(G, C, device=dev, dtype=torch.int32)
        check(lib.papc_bn_relu_max_f32(ptr(x), ptr(one), ptr(zero), G, K, C, ptr(out), ptr(argmax), stream_ptr()), "papc_bn_relu_max_f32")
        ctx.save_for_backward(argmax)
        ctx.K = K
        return out

    @staticmethod
    def backward(ctx, gout):
        (argmax,) = ctx.saved_tensors
        G, C = argmax.shape
        gout = gout.contiguous().float()
        dx = torch.empty(G * ctx.K, C, device=gout.device, dtype=torch.float32)
        check(_lib.load().papc_group_max_bwd_f32(ptr(gout), ptr(argmax), G, ctx.K, C, ptr(dx), stream_ptr()), "papc_group_max_bwd_f32")
        return dx, None, None


class _LinearReLU(torch.autograd.Function):
    """relu(rows @ w^T + b) without a norm (PFNLayer(use_norm=False): Linear with bias + the Empty norm, pillars.py:25-27,:30-32) on the
    MFMA row kernels.  The backward reuses the BN-aware kernels with identity constants (scale 1, shift 0, mean 0, invstd 1,
    c1 = c2 = 0): their dY is then exactly gout * [y > 0]."""

    @staticmethod
    def forward(ctx, rows, w, b):
        lib = _lib.load()
        st = stream_ptr()
        M, cin = rows.shape
        cout = w.shape[0]
        dev = rows.device
        y = torch.empty(M, cout, device=dev, dtype=torch.float32)
        check(lib.papc_mlp_gemm_f32(0, ptr(rows), cin, None, None, None, ptr(w), ptr(b), M, cin, cout, ptr(y), None, None, st),
              "papc_mlp_gemm_f32")
        ident = torch.zeros(4, cout, device=dev, dtype=torch.float32)    # mean 0, invstd / scale 1, shift 0
        ident[1].fill_(1.0)
        ident[2].fill_(1.0)
        x = torch.empty(M, cout, device=dev, dtype=torch.float32)
        check(lib.papc_bn_relu_f32(ptr(y), ident[2].data_ptr(), ident[3].data_ptr(), M, cout, ptr(x), st), "papc_bn_relu_f32")
        ctx.save_for_backward(rows, w, y, ident)
        ctx.needs = (rows.requires_grad, )
        return x

    @staticmethod
    def backward(ctx, gout):
        from ._lib import BwdDy
        from .mlp import _dw_rows_per_chunk
        import ctypes
        lib = _lib.load()
        st = stream_ptr()
        rows, w, y, ident = ctx.saved_tensors
        M, cin = rows.shape
        cout = w.shape[0]
        dev = rows.device
        gout = gout.contiguous().float()
        zero2 = torch.zeros(2, cout, device=dev, dtype=torch.float32)
        dy = BwdDy()
        dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = 0, gout.data_ptr(), None, None, 1
        dy.y = y.data_ptr()
        dy.mean, dy.invstd, dy.scale, dy.shift = ident[0].data_ptr(), ident[1].data_ptr(), ident[2].data_ptr(), ident[3].data_ptr()
        dy.c1, dy.c2 = zero2[0].data_ptr(), zero2[1].data_ptr()
        rpc = _dw_rows_per_chunk(M, cout, cin)
        n_chunks = (M + rpc - 1) // rpc
        pld = cout * cin + cout
        part = torch.empty(n_chunks, pld, device=dev, dtype=torch.float32)
        check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dy), 0, ptr(rows), cin, None, None, None, M, cin, cout, rpc, part.data_ptr(),
                                      part.data_ptr() + 4 * cout * cin, pld, st), "papc_mlp_bwd_dw_f32")
        dw = torch.empty(cout, cin, device=dev, dtype=torch.float32)
        dbz = torch.empty(cout, device=dev, dtype=torch.float32)
        check(lib.papc_reduce_partials2_f32(ptr(part), n_chunks, pld, cout * cin, ptr(dw), cout, ptr(dbz), 0, st), "papc_reduce_partials2_f32")
        # the dW kernel writes the bias gradient of a BN-fed conv (exactly 0); without a norm it is the column sum of dY:
        # (sum p, sum p*xhat) from the BN-backward reduce with the identity constants -> its first row
        n_parts = min(512, (M + 127) // 128)
        red = torch.empty(n_parts, 2, cout, device=dev, dtype=torch.float32)
        check(lib.papc_bn_bwd_reduce_f32(0, ptr(gout), None, None, 1, ptr(y), ident[0].data_ptr(), ident[1].data_ptr(), ident[2].data_ptr(),
                                         ident[3].data_ptr(), M, cout, n_parts, ptr(red), st), "papc_bn_bwd_reduce_f32")
        dgb = torch.empty(2, cout, device=dev, dtype=torch.float32)
        c12 = torch.empty(2, cout, device=dev, dtype=torch.float32)
        check(lib.papc_bn_bwd_finalize_f32(ptr(red), n_parts, M, cout, dgb[0].data_ptr(), dgb[1].data_ptr(), c12[0].data_ptr(),
                                           c12[1].data_ptr(), 2, st), "papc_bn_bwd_finalize_f32")
        db = dgb[1]
        dx = None
        if ctx.needs[0]:
            wt = w.t().contiguous()
            dx = torch.empty(M, cin, device=dev, dtype=torch.float32)
            check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, cin, cout, ptr(dx), None, None, st), "papc_mlp_bwd_dx_f32")
        return dx, dw, db


class PFNLayer(nn.Module):
    """pillars.py:9-41."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.name = 'PFNLayer'
        self.last_vfe = last_layer
        if not self.last_vfe:
            out_channels = out_channels // 2                                                   # :18-19
        self.units = out_channels
        self.use_norm = bool(use_norm)
        if use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)                     # :23
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)                  # :24 (paddle momentum)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)                      # :26
            self.norm = nn.Identity()                                                          # :27 (libs.nn.Empty)

    def activations(self, rows):
        """relu(norm(linear(rows))) for rows [M, Cin] -> [M, C] (:30-32), differentiable."""
        C = self.units
        if not self.use_norm:
            return _LinearReLU.apply(rows, self.linear.weight, self.linear.bias)
        M = rows.shape[0]
        spec = StackSpec(1, M, M, 1, 0, True, eps=self.norm.eps, momentum=self.norm.momentum, pool=False,
                         eval_bn=not self.training)
        zero_b = torch.zeros(C, device=rows.device, dtype=torch.float32)
        return shared_mlp_max(spec, [(self.norm.running_mean, self.norm.running_var)], None, None, None, None,
                              [self.linear.weight, zero_b, self.norm.weight, self.norm.bias], x_rows=rows)

    def forward(self, inputs):
        """inputs [P,T,Cin] -> [P,1,C] (last layer, :36-37) or [P,T,2C] (concat with the tiled max, :39-41)."""
        P, T, Cin = inputs.shape
        C = self.units
        rows = inputs.reshape(P * T, Cin).contiguous().float()
        if self.last_vfe and self.use_norm:
            # the max is the only output: one fused node, the [P*T, C] activations are never materialised
            spec = StackSpec(P, T, 1, T, 0, True, eps=self.norm.eps, momentum=self.norm.momentum, eval_bn=not self.training)
            zero_b = torch.zeros(C, device=rows.device, dtype=torch.float32)
            out = shared_mlp_max(spec, [(self.norm.running_mean, self.norm.running_var)], None, None, None, None,
                                 [self.linear.weight, zero_b, self.norm.weight, self.norm.bias], x_rows=rows)
            return out.view(P, 1, C)
        x = self.activations(rows)                                                             # [P*T, C]
        x_max = _GroupMax.apply(x, P, T).view(P, 1, C)                                         # :34
        if self.last_vfe:
            return x_max                                                                       # :36-37
        return torch.cat([x.view(P, T, C), x_max.expand(P, T, C)], dim=2)                      # :39-41


class PillarFeatureNet(nn.Module):
    """pillars.py:43-108."""

    def __init__(self, num_input_features=4, use_norm=True, num_filters=(64, 128), with_distance=False,
                 voxel_size=(0.2, 0.2, 4), pc_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        self.name = 'PillarFeatureNet'
        assert len(num_filters) > 0
        if num_input_features < 3:
            raise ValueError("PillarFeatureNet: points are [x, y, z, ...] (num_input_features >= 3; :82-88 read the first three columns)")
        self._nf = int(num_input_features)          # (4 = the reference's voxeliser, yaml NUM_POINT_FEATURES: 4 -- the fused kernels' layout)
        num_input_features += 5                                                                # :55
        if with_distance:
            num_input_features += 1                                                            # :57-58
        self._with_distance = with_distance
        self._use_norm = bool(use_norm)
        num_filters = [num_input_features] + list(num_filters)
        layers = []
        for i in range(len(num_filters) - 1):
            layers.append(PFNLayer(num_filters[i], num_filters[i + 1], use_norm, last_layer=(i >= len(num_filters) - 2)))
        self.pfn_layers = nn.ModuleList(layers)                                                # :71
        # the fused path's two ticket words (include/papc_hip.h: papc_pfn_io.tickets): owned by this module, zero between launches; a module
        # runs on one stream at a time, so this is the per-stream pair the header asks for (no process-global state in the library)
        self.register_buffer("_tickets", torch.zeros(16, dtype=torch.int32), persistent=False)
        # True: the caller guarantees zero rows behind num_voxels (what the reference's voxeliser produces, point_cloud_ops.py:148) ->
        # the kernels load only the real rows.  Off by default: with other padding the cluster mean (:82) would differ
        self.assume_zero_padding = False
        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]                                              # :76
        self.y_offset = self.vy / 2 + pc_range[1]                                              # :77

    def _geom(self):
        return (float(self.vx), float(self.vy), float(self.x_offset), float(self.y_offset))

    def decorate(self, features, num_voxels, coors):
        """:82-102 only -> masked rows [P,T,F+5] (+ the point norm as one more channel with_distance, :92-94).  The inputs are data
        (no gradient flows into the point cloud)."""
        lib = _lib.load()
        features = features.contiguous().float()
        P, T, F = features.shape
        if F != self._nf:
            raise ValueError("PillarFeatureNet: points of width %d, built for num_input_features=%d" % (F, self._nf))
        nc = F + 5 + (1 if self._with_distance else 0)
        out = torch.empty(P, T, nc, device=features.device, dtype=torch.float32)
        vx, vy, xo, yo = self._geom()
        if F != 4 or T > 128:      # (the F = 4 kernel keeps a pillar's <= 128 points in two float4 registers per lane)
            check(lib.papc_pfn_decorate_nf_f32(ptr(features), F, ptr(num_voxels.int().contiguous()), ptr(coors.int().contiguous()), P, T,
                                               vx, vy, xo, yo, int(self._with_distance), ptr(out), stream_ptr()), "papc_pfn_decorate_nf_f32")
            return out
        check(lib.papc_pfn_decorate_f32(ptr(features), ptr(num_voxels.int().contiguous()), ptr(coors.int().contiguous()), P, T,
                                        vx, vy, xo, yo, int(self._with_distance), ptr(out), stream_ptr()), "papc_pfn_decorate_f32")
        return out

    def forward(self, features, num_voxels, coors):
        """features [P,T,F] f32 (F = num_input_features, 4 in the reference's configs), num_voxels [P] int, coors [P,4] int
        (batch,z,y,x) -> [P,C]."""
        if not features.is_cuda:
            raise _lib.PapcError("PillarFeatureNet needs CUDA(ROCm) tensors (no CPU fallback)")
        features = features.contiguous().float()
        num_voxels = num_voxels.int().contiguous()
        coors = coors.int().contiguous()
        pfn = self.pfn_layers[0]
        if self._nf == 4 and features.shape[1] <= 128 and len(self.pfn_layers) == 1 and self._use_norm and not self._with_distance and pfn.units <= 64:
            out = _PFNFused.apply(self._geom(), features, num_voxels, coors, pfn.linear.weight, pfn.norm.weight, pfn.norm.bias,
                                  pfn.norm.running_mean, pfn.norm.running_var, pfn.norm.eps, pfn.norm.momentum, self.training,
                                  self._tickets, self.assume_zero_padding)
            return out.squeeze()                                                               # :108
        with torch.no_grad():
            x = self.decorate(features, num_voxels, coors)
        for pfn in self.pfn_layers:                                                            # :105-106
            x = pfn(x)
        return x.squeeze()
