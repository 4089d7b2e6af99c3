"""The shared pointwise MLP stack through the library's coarse entry points: ONE C call per direction
(include/papc_hip.h: papc_sa_mlp_plan / papc_sa_mlp_fwd / papc_sa_mlp_bwd, csrc/sa_mlp.hip).

Reference: /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:214-219 (SSG), :271-276 (MSG), :330-333 (feature propagation),
/root/reference/PAPC/models/classify/pointnet_base/pointnet_base.py:7-25,44.  The path per layer (gather-add first layer, coordinates-only
first layer through its moments, row-streaming / tiled GEMMs, fused max, no-store max layer, compacted stack) and every scratch buffer
are the library's business; this module only marshals pointers: three device buffers (saved, forward scratch, backward scratch), the
parameters, and the gradient targets.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

MAXL = 8
c_p, c_i, c_l, c_f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
NO_LINGATHER, NO_XYZ1, NO_NOSTORE, NO_GMAX, NO_FUSED_RED, NO_COMPACT, NO_PLANES, NO_PLANES_POINTWISE, NO_XYZ_FUSE, NO_WSTATS, NO_PSEL, NO_GSIGN = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048


class SaDesc(ctypes.Structure):
    """papc_sa_desc"""
    _fields_ = [("B", c_i), ("N", c_i), ("S", c_i), ("K", c_i), ("D", c_i), ("n_layers", c_i), ("cin", c_i), ("cout", c_i * MAXL),
                ("input", c_i), ("identity_rows", c_i), ("xyz_first", c_i), ("pool", c_i), ("eval_bn", c_i), ("cut_gather_grad", c_i),
                ("eps", c_f), ("momentum", c_f), ("disable", ctypes.c_uint32), ("inference", c_i), ("want_input_grad", c_i)]


class SaLayer(ctypes.Structure):
    """papc_sa_layer"""
    _fields_ = [("w", c_p), ("b", c_p), ("gamma", c_p), ("beta", c_p), ("running_mean", c_p), ("running_var", c_p)]


class CompactSrc(ctypes.Structure):
    """papc_compact_src"""
    _fields_ = [("start", c_p), ("rows", c_p), ("cidx", c_p), ("seg_grp", c_p), ("wrow", c_p), ("coef", c_p), ("G", c_i)]


class SaIo(ctypes.Structure):
    """papc_sa_io"""
    _fields_ = [("xyz", c_p), ("sb", c_l), ("sn", c_l), ("sc", c_l), ("new_xyz", c_p), ("feats", c_p), ("idx", c_p), ("x_rows", c_p),
                ("xc", c_p), ("xc_gram", c_p), ("compact", ctypes.POINTER(CompactSrc)), ("consts3", c_p), ("consts3_ld", c_i),
                ("layer", SaLayer * MAXL), ("out", c_p), ("saved", c_p), ("scratch", c_p), ("plists", c_p), ("wfeat", c_p)]


class SaPlan(ctypes.Structure):
    """papc_sa_plan"""
    _fields_ = [("d", SaDesc), ("cin0", c_i), ("lin0", c_i), ("xyz1", c_i), ("gmax", c_i), ("nostore", c_i), ("compact", c_i), ("sparse_max", c_i),
                ("planes", c_i), ("saved_bytes", c_l), ("fwd_scratch_bytes", c_l), ("bwd_scratch_bytes", c_l), ("off_y", c_l * MAXL), ("off_cst", c_l * MAXL),
                ("off_argmax", c_l)]


class SaGrads(ctypes.Structure):
    """papc_sa_grads"""
    _fields_ = [("gout", c_p), ("dw", c_p * MAXL), ("db", c_p * MAXL), ("dgamma", c_p * MAXL), ("dbeta", c_p * MAXL), ("acc_w", c_i * MAXL),
                ("acc_gb", c_i * MAXL), ("wt", c_p * MAXL), ("grad_feats", c_p), ("grad_x", c_p), ("defer", c_p)]


_CONSTS3 = {}
LAST_PLANS = {}


def _consts3(dev):
    """ones | zeros | 1e30, three rows of 1024 floats: the identity BatchNorm constants the gather-add backward hands the dW kernel"""
    t = _CONSTS3.get(str(dev))
    if t is None:
        mk = lambda: torch.tensor([1.0, 0.0, 1e30], device=dev, dtype=torch.float32).view(3, 1).expand(3, 1024).contiguous()
        if _lib._capturing():
            return mk()
        t = _CONSTS3[str(dev)] = mk()
    return t


def _disable_bits():
    from . import mlp
    bits = 0
    if not mlp._LIN_GATHER: bits |= NO_LINGATHER
    if not mlp._XYZ1: bits |= NO_XYZ1
    if not mlp._NOSTORE: bits |= NO_NOSTORE
    if not mlp._FUSE_GMAX: bits |= NO_GMAX
    if not mlp._FUSE_RED: bits |= NO_FUSED_RED
    from . import smallm
    if not smallm.ENABLED: bits |= NO_PLANES
    if not smallm.POINTWISE: bits |= NO_PLANES_POINTWISE
    if not mlp._XYZ_FUSE: bits |= NO_XYZ_FUSE
    if os.environ.get("PAPC_GSIGN", "1") == "0": bits |= NO_GSIGN       # A/B: both extrema per channel in the fused group max (no sign(gamma) hint)
    if os.environ.get("PAPC_PSEL", "1") == "0": bits |= NO_PSEL         # A/B: the compacted max layer's dX from gout (ReLU test + scale per row)
    if os.environ.get("PAPC_WSTATS", "1") == "0": bits |= NO_WSTATS     # A/B: unweighted statistics + papc_bn_stats_corr_f32 per layer
    return bits


def _fill_io(io, spec, xyz, new_xyz, feats, idx, x_rows, params, bn_buffers, keep):
    """pointers of the inputs and parameters; ``keep`` collects objects that must outlive the call"""
    L = len(params) // 4
    if x_rows is None:
        io.xyz = xyz.data_ptr()
        io.sb, io.sn, io.sc = xyz.stride(0), xyz.stride(1), xyz.stride(2)
        io.new_xyz, io.feats, io.idx = new_xyz.data_ptr(), ptr(feats), ptr(idx)
    else:
        io.x_rows = x_rows.data_ptr()
    if spec.xyz_pre is not None:
        io.xc, io.xc_gram = spec.xyz_pre[0].data_ptr(), spec.xyz_pre[1].data_ptr()
    cp = spec.compact
    if cp is not None:
        src = CompactSrc(cp.start.data_ptr(), cp.rows.data_ptr(), cp.cidx.data_ptr(), cp.seg_grp.data_ptr(), cp.wrow.data_ptr(), cp.coef.data_ptr(), cp.G)
        keep.append(src)
        io.compact = ctypes.pointer(src)
    pl = getattr(spec, "plists", None)
    if pl is not None:          # the grouping's point lists (compact.point_lists): deterministic gather-add backward
        src = _lib.PointLists(pl.prange.data_ptr(), pl.prow.data_ptr(), pl.pmeta.data_ptr(), int(pl.compact),
                              pl.pmom.data_ptr() if getattr(pl, "pmom", None) is not None else None)
        keep.append(src)
        keep.append(pl)
        io.plists = ctypes.addressof(src)
    wtab = getattr(spec, "wt_table", None)
    if wtab is not None and feats is not None and L >= 1:
        # the first layer's feature block, made contiguous for THIS forward by mlp.precompute_wt(feat_blocks=...): shape-checked, like the transposes
        ent = wtab.get(("f", params[0].data_ptr()))
        cin0 = params[0].numel() // params[0].shape[0]
        if ent is not None and ent[1] == bool(spec.xyz_first) and tuple(ent[0].shape) == (params[0].shape[0], cin0 - 3) and cin0 - 3 == spec.D:
            io.wfeat = ent[0].data_ptr()
            keep.append(ent[0])
    c3 = _consts3(params[0].device)
    keep.append(c3)
    io.consts3, io.consts3_ld = c3.data_ptr(), c3.shape[1]
    for l in range(L):
        w, b, gamma, beta = params[4 * l: 4 * l + 4]
        ly = io.layer[l]
        assert w.is_contiguous()
        ly.w, ly.b, ly.gamma, ly.beta = w.data_ptr(), ptr(b), gamma.data_ptr(), beta.data_ptr()
        if bn_buffers is not None:
            ly.running_mean, ly.running_var = ptr(bn_buffers[l][0]), ptr(bn_buffers[l][1])


class SharedMLPStack(torch.autograd.Function):
    """out[g, :] = max_{k<K} relu(bn_L(conv_L(... relu(bn_1(conv_1(rows)))...)))  -- same arguments and semantics as mlp.SharedMLPMax, one
    library call per direction.  apply(spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, w1, b1, gamma1, beta1, w2, ...)"""

    @staticmethod
    def forward(ctx, spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, *params):
        lib = _lib.load()
        dev = params[0].device
        L = len(params) // 4
        assert 1 <= L <= MAXL
        plain = x_rows is not None
        d = SaDesc()
        d.B, d.N, d.S, d.K, d.D, d.n_layers = spec.B, spec.N, spec.S, spec.K, spec.D, L
        d.cin = x_rows.shape[1] if plain else spec.D + 3
        for l in range(L):
            d.cout[l] = params[4 * l].shape[0]
        d.input = 1 if plain else 0
        d.identity_rows = int((not plain) and idx is None)
        d.xyz_first, d.pool, d.eval_bn, d.cut_gather_grad = int(spec.xyz_first), int(spec.pool), int(spec.eval_bn), int(bool(spec.cut_gather_grad))
        d.eps, d.momentum = spec.eps, spec.momentum
        d.disable = _disable_bits()
        if feats is not None and not feats.is_contiguous():
            d.disable |= NO_LINGATHER | NO_PLANES
        if plain and not x_rows.is_contiguous():
            d.disable |= NO_PLANES
        d.inference = int(not any(ctx.needs_input_grad))
        d.want_input_grad = int((plain and x_rows.requires_grad) or ((not plain) and feats is not None and feats.requires_grad and not spec.cut_gather_grad))
        keep = []
        io = SaIo()
        _fill_io(io, spec, xyz, new_xyz, feats, idx, x_rows, params, bn_buffers, keep)
        plan = SaPlan()
        check(lib.papc_sa_mlp_plan(ctypes.byref(d), ctypes.byref(io), ctypes.byref(plan)), "papc_sa_mlp_plan")
        M = spec.M
        G = spec.B * spec.S
        cL = params[4 * (L - 1)].shape[0]
        out = torch.empty(G if spec.pool else M, cL, device=dev, dtype=torch.float32)
        saved = torch.empty(plan.saved_bytes, device=dev, dtype=torch.uint8)
        scratch = torch.empty(plan.fwd_scratch_bytes, device=dev, dtype=torch.uint8)
        io.out, io.saved, io.scratch = out.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        check(lib.papc_sa_mlp_fwd(ctypes.byref(plan), ctypes.byref(io), stream_ptr()), "papc_sa_mlp_fwd")
        ctx.spec, ctx.L, ctx.plan, ctx.saved = spec, L, plan, saved
        # (diagnostics: which path the library chose for a stack of this shape -- bench.py prices its roofline line on the tensors these
        # paths really move)
        LAST_PLANS[(int(M), tuple(int(d.cout[l]) for l in range(L)))] = dict(
            lin0=bool(plan.lin0), xyz1=bool(plan.xyz1), gmax=bool(plan.gmax), nostore=bool(plan.nostore), compact=bool(plan.compact), planes=bool(plan.planes),
            xyz_fuse=bool(plan.xyz1) and not (d.disable & NO_XYZ_FUSE))
        ctx.compact = spec.compact if plan.compact else None
        ctx.nostore, ctx.xyz1, ctx.lin0, ctx.planes = bool(plan.nostore), bool(plan.xyz1), bool(plan.lin0), bool(plan.planes)     # (which paths the library took)
        ctx.bn_buffers = bn_buffers
        ctx.wt_table = spec.wt_table
        ctx.feats_needs_grad = feats is not None and feats.requires_grad and not spec.cut_gather_grad
        ctx.x_needs_grad = plain and x_rows.requires_grad
        ctx.save_for_backward(xyz, new_xyz, feats, idx, x_rows, *params)
        return out

    # ---- what the forward left behind, as views into the saved buffer (tests read the kernels' decisions from them)
    @staticmethod
    def views(ctx):
        """(argmax [G, c_L] int32 or None, [y_l or None], [cst_l [4, c_l]])"""
        plan, saved, spec, L = ctx.plan, ctx.saved, ctx.spec, ctx.L
        M, G = spec.M, spec.B * spec.S
        def view(off, n, dt):
            return None if off < 0 else saved[off: off + 4 * n].view(dt)
        cs = [plan.d.cout[l] for l in range(L)]
        am = view(plan.off_argmax, G * cs[-1], torch.int32)
        ys = [None if plan.off_y[l] < 0 else view(plan.off_y[l], M * cs[l], torch.float32).view(M, cs[l]) for l in range(L)]
        cst = [view(plan.off_cst[l], 4 * cs[l], torch.float32).view(4, cs[l]) for l in range(L)]
        return (None if am is None else am.view(G, cs[-1])), ys, cst

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        spec, L, plan = ctx.spec, ctx.L, ctx.plan
        tens = ctx.saved_tensors
        xyz, new_xyz, feats, idx, x_rows = tens[:5]
        params = tens[5:]
        dev = gout.device
        gout = gout.contiguous().float()
        keep = []
        io = SaIo()
        _fill_io(io, spec, xyz, new_xyz, feats, idx, x_rows, params, ctx.bn_buffers, keep)
        scratch = torch.empty(plan.bwd_scratch_bytes, device=dev, dtype=torch.uint8)
        io.saved, io.scratch = ctx.saved.data_ptr(), scratch.data_ptr()
        g = SaGrads()
        g.gout = gout.data_ptr()
        grads = [None] * (4 * L)
        all_inplace = True
        for l in range(L):
            w = params[4 * l]
            cout = w.shape[0]
            cin = w.numel() // cout
            tgt = spec.grad_targets[4 * l: 4 * l + 4] if spec.grad_targets is not None else None
            inplace = tgt is not None and all(t is not None for t in tgt)
            gb_inplace = tgt is not None and tgt[2] is not None and tgt[3] is not None and not spec.eval_bn
            all_inplace = all_inplace and inplace
            if inplace:          # accumulate straight into the parameters' .grad (flat-bucket views): no autograd add kernels
                g.dw[l], g.db[l], g.acc_w[l] = tgt[0].data_ptr(), tgt[1].data_ptr(), 1
            else:
                dw = torch.empty(cout, cin, device=dev, dtype=torch.float32)
                grads[4 * l] = dw.reshape(w.shape)
                g.dw[l], g.acc_w[l] = dw.data_ptr(), 0
                if tgt is not None and tgt[1] is not None and not spec.eval_bn:
                    g.db[l] = None                      # (a bias under a train-mode BN: gradient exactly 0 -- nothing to add in place)
                else:
                    db = torch.empty(cout, device=dev, dtype=torch.float32)
                    grads[4 * l + 1] = db
                    g.db[l] = db.data_ptr()
            if gb_inplace and not (l == 0 and plan.xyz1 and not inplace):
                g.dgamma[l], g.dbeta[l], g.acc_gb[l] = tgt[2].data_ptr(), tgt[3].data_ptr(), 1
            else:                # (the coordinates-only first layer takes one accumulate flag for its three outputs)
                dgb = torch.empty(2, cout, device=dev, dtype=torch.float32)
                grads[4 * l + 2], grads[4 * l + 3] = dgb[0], dgb[1]
                g.dgamma[l], g.dbeta[l], g.acc_gb[l] = dgb[0].data_ptr(), dgb[1].data_ptr(), 0
            if ctx.wt_table is not None:
                t = ctx.wt_table.get(w.data_ptr())
                if t is not None and tuple(t.shape) == (cin, cout):
                    g.wt[l] = t.data_ptr()
        grad_feats = grad_x = None
        if ctx.feats_needs_grad:
            grad_feats = torch.empty(spec.B, spec.N, spec.D, device=dev, dtype=torch.float32)
            g.grad_feats = grad_feats.data_ptr()
        if ctx.x_needs_grad:
            grad_x = torch.empty(spec.M, x_rows.shape[1], device=dev, dtype=torch.float32)
            g.grad_x = grad_x.data_ptr()
        # the partial folds of this stack join the running backward pass's list (folds.py: one launch behind the last backward kernel)
        # (only when every weight gradient lands in place in its .grad: a tensor handed back to autograd must be complete on return)
        from . import folds
        fl = folds.pending([scratch, ctx.saved]) if all_inplace else None
        if fl is not None:
            g.defer = ctypes.cast(fl, c_p)
        check(lib.papc_sa_mlp_bwd(ctypes.byref(plan), ctypes.byref(io), ctypes.byref(g), stream_ptr()), "papc_sa_mlp_bwd")
        return (None, None, None, None, grad_feats, None, grad_x) + tuple(grads)
