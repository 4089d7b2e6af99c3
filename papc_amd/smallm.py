"""Shared MLP stack + max on FEW rows (the group_all set-abstraction layer) through the "planes" kernels of csrc/smallm.hip.

Reference: /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:160-176 (sample_and_group_all: rows [xyz | feats], centre 0),
:215-219 (relu(bn(conv(.))) x L, max over the group); PointNet2_SSG_Clas.sa3 (classify/pointnet2/pointnet2.py:16), the MSG and
part-segmentation nets' sa3.

Same contract as mlp.SharedMLPMax (one autograd node per stack, only the pre-BN outputs y_l kept) but a different decomposition,
built for M = B*128 rows where the row-GEMM kernels are latency-bound: every operand is written ONCE as fragment-ordered bf16
planes by a prep kernel that also folds the train-mode BatchNorm statistics (forward) or the two BN-backward constants (backward)
from the producer's per-tile partials, and every product of the stack -- forward, dX, dW -- is the same plane-set GEMM
(include/papc_hip.h: papc_pg_*).  Launches per step: 8 forward, 10 backward (the row kernels needed 7 + 13).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

PLAIN, CONCAT, BNRELU, DY_DENSE, DY_MAX = 0, 1, 2, 3, 4
DZ_DENSE = 0                                            # papc_bwd_dy.dz_mode of a dense upstream gradient (PAPC_DZ_DENSE)
EPI_STORE, EPI_FWD, EPI_FWD_GMAX, EPI_RED = 0, 1, 2, 3
K_MLP_GEMM, K_BWD_DX, K_BWD_DW = 3, 6, 7                # PAPC_K_* families for the event profiler
GROUP = 128                                             # rows of a group = rows of a GEMM tile (fused max)
ENABLED = os.environ.get("PAPC_PLANES", "1") == "1"     # A/B switch: 0 = every stack on the row kernels (mlp.SharedMLPMax)
MAX_ROWS = int(os.environ.get("PAPC_PLANES_MAXROWS", "16384"))
POINTWISE = os.environ.get("PAPC_PLANES_POINTWISE", "1") == "1"   # A/B switch: stacks without pooling (feature propagation) on the planes path too

c_p, c_i, c_l, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class PgWJob(ctypes.Structure):
    """papc_pg_wjob"""
    _fields_ = [("src", c_p), ("row_stride", c_l), ("col_stride", c_l), ("R", c_i), ("K", c_i), ("planes", c_p)]


class PgPrep(ctypes.Structure):
    """papc_pg_prep"""
    _fields_ = [("mode", c_i), ("M", c_l), ("C", c_i), ("x", c_p), ("ldx", c_l), ("xyz", c_p), ("sb", c_l), ("sn", c_l), ("sc", c_l),
                ("feats", c_p), ("N", c_i), ("D", c_i), ("xyz_first", c_i), ("dz", c_p), ("gout", c_p), ("ysel", c_p), ("argmax", c_p),
                ("K", c_i), ("stats", c_p), ("parts", c_i), ("gamma", c_p), ("beta", c_p), ("eps", c_f), ("momentum", c_f),
                ("running_mean", c_p), ("running_var", c_p), ("mean", c_p), ("invstd", c_p), ("scale", c_p), ("shift", c_p),
                ("red", c_p), ("red_parts", c_i), ("dgamma", c_p), ("dbeta", c_p), ("accumulate", c_i), ("planes", c_p),
                ("planes_t", c_p)]


class PgGemm(ctypes.Structure):
    """papc_pg_gemm"""
    _fields_ = [("epi", c_i), ("a", c_p), ("b", c_p), ("R1", c_i), ("R2", c_i), ("K", c_i), ("c", c_p), ("ldc", c_l), ("split", c_i),
                ("split_stride", c_l), ("bias", c_p), ("stats", c_p), ("gmax", c_p), ("gmin", c_p), ("amax", c_p), ("amin", c_p),
                ("y_prev", c_p), ("mean", c_p), ("invstd", c_p), ("scale", c_p), ("shift", c_p), ("family", c_i)]


class PgFoldJob(ctypes.Structure):
    """papc_pg_fold_job"""
    _fields_ = [("partial", c_p), ("nsplit", c_i), ("stride", c_l), ("n", c_l), ("out", c_p), ("accumulate", c_i)]


def eligible(spec, xyz, feats, idx, x_rows, params):
    """The stack shapes smallm.hip was built for: train-mode BN, max over groups of exactly 128 rows (one GEMM tile), no neighbour
    index (group_all or plain rows), few rows, channel counts in multiples of 8."""
    if not ENABLED or spec.eval_bn or idx is not None:
        return False
    M = spec.M
    if M % GROUP or M > MAX_ROWS or M < GROUP:
        return False
    if spec.pool and spec.K != GROUP:
        return False
    if not spec.pool and (x_rows is None or not POINTWISE):   # (no pooling: the point-wise stacks of feature propagation, plain rows only)
        return False
    if len(params) < 8:                                    # (a single layer gains nothing here)
        return False
    couts = [params[4 * l].shape[0] for l in range(len(params) // 4)]
    if any(c % 8 for c in couts):
        return False
    if x_rows is not None:
        return x_rows.is_cuda and x_rows.is_contiguous() and x_rows.shape[1] % 8 == 0 and x_rows.dtype == torch.float32
    if xyz is None or not xyz.is_cuda or spec.S != 1:
        return False
    return feats is None or (feats.is_contiguous() and feats.dtype == torch.float32)


def takes_group_all(B, n_points, couts):
    """eligible() for a sample_and_group_all stack of ``B`` clouds of ``n_points`` points, from shapes alone (train mode, contiguous fp32 CUDA
    inputs assumed): lets a model skip work the planes path does not use (the W^T table of mlp.precompute_wt)."""
    M = B * n_points
    return bool(ENABLED and n_points == GROUP and M <= MAX_ROWS and len(couts) >= 2 and not any(c % 8 for c in couts))


def _planes(lib, R, K, dev):
    return torch.empty(lib.papc_pg_planes_bytes(R, K), dtype=torch.uint8, device=dev)


def _split_for(R1, R2, nst):
    """split-K factor of a dW product: about one workgroup per CU, at least 8 k32 stages each (2 for products of one or two output tiles), a power
    of two dividing nst (sa_mlp.hip::pg_split_for: the same rule)"""
    tiles = ((R1 + 127) // 128) * ((R2 + 127) // 128)
    s = 1
    min_stages = 2 if tiles <= 2 else 8
    while s * 2 * tiles <= 256 and nst % (s * 2) == 0 and nst // (s * 2) >= min_stages:
        s *= 2
    return s


class PlanesMLPMax(torch.autograd.Function):
    """apply(spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, w1, b1, gamma1, beta1, ...) -- the arguments of mlp.SharedMLPMax
    (idx must be None; new_xyz is the zero centre of sample_and_group_all and is not read)."""

    @staticmethod
    def forward(ctx, spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, *params):
        lib = _lib.load()
        st = stream_ptr()
        dev = params[0].device
        L = len(params) // 4
        M = spec.M
        T = M // GROUP                                       # row tiles = groups
        plain = x_rows is not None
        cin0 = x_rows.shape[1] if plain else spec.D + 3
        ch = [cin0] + [params[4 * l].shape[0] for l in range(L)]
        want_bwd = any(ctx.needs_input_grad)
        in_grad = (plain and x_rows.requires_grad) or ((not plain) and feats is not None and feats.requires_grad and not spec.cut_gather_grad)
        fcol0 = (3 if spec.xyz_first else 0) if not plain else 0
        n_in = cin0 if plain else spec.D                      # input columns that carry a gradient

        # ---- weights -> planes: W_l [c_l x c_(l-1)] for the forward, W_l^T [c_(l-1) x c_l] for dX (layer 1: the gradient-carrying columns)
        w2s, wp, wtp, jobs = [], [], [None] * L, []
        for l in range(L):
            w2 = params[4 * l].reshape(ch[l + 1], ch[l])
            assert w2.is_contiguous()
            w2s.append(w2)
            wp.append(_planes(lib, ch[l + 1], ch[l], dev))
            jobs.append((w2.data_ptr(), ch[l], 1, ch[l + 1], ch[l], wp[l].data_ptr()))
            if want_bwd and (l > 0 or in_grad):
                rows, off = (ch[l], 0) if l > 0 else (n_in, fcol0)
                wtp[l] = _planes(lib, rows, ch[l + 1], dev)
                jobs.append((w2.data_ptr() + 4 * off, 1, ch[l], rows, ch[l + 1], wtp[l].data_ptr()))
        for j0 in range(0, len(jobs), 8):
            chunk = jobs[j0:j0 + 8]
            arr = (PgWJob * len(chunk))()
            for a, (src, sr, sc, R, K, dst) in zip(arr, chunk):
                a.src, a.row_stride, a.col_stride, a.R, a.K, a.planes = src, sr, sc, R, K, dst
            check(lib.papc_pg_prep_weights_f32(arr, len(chunk), st), "papc_pg_prep_weights_f32")

        # ---- layer 1 operand: the rows of sample_and_group_all (or the caller's rows)
        def prep(**kw):
            a = PgPrep()
            for k, v in kw.items():
                setattr(a, k, v)
            check(lib.papc_pg_prep_rows_f32(ctypes.byref(a), st), "papc_pg_prep_rows_f32")

        P = _planes(lib, M, cin0, dev)
        PT = [None] * L                                        # input^T planes of every layer (the dW operands)
        if want_bwd:
            PT[0] = _planes(lib, cin0, M, dev)
        if plain:
            prep(mode=PLAIN, M=M, C=cin0, x=x_rows.data_ptr(), ldx=cin0, planes=P.data_ptr(), planes_t=ptr(PT[0]))
        else:
            prep(mode=CONCAT, M=M, C=cin0, xyz=xyz.data_ptr(), sb=xyz.stride(0), sn=xyz.stride(1), sc=xyz.stride(2), feats=ptr(feats),
                 N=spec.N, D=spec.D, xyz_first=int(spec.xyz_first), planes=P.data_ptr(), planes_t=ptr(PT[0]))
        ys, consts = [], []
        gbuf_f = gbuf_i = None
        for l in range(L):
            cout = ch[l + 1]
            y = torch.empty(M, cout, device=dev, dtype=torch.float32)
            stats = torch.empty(T, 2, cout, device=dev, dtype=torch.float32)
            cst = torch.empty(4, cout, device=dev, dtype=torch.float32)   # mean, invstd, scale, shift
            g = PgGemm()
            g.epi = EPI_FWD_GMAX if (l == L - 1 and spec.pool) else EPI_FWD
            g.a, g.b, g.R1, g.R2, g.K = P.data_ptr(), wp[l].data_ptr(), M, cout, ch[l]
            g.c, g.ldc, g.split, g.split_stride = y.data_ptr(), cout, 1, 0
            g.bias, g.stats, g.family = ptr(params[4 * l + 1]), stats.data_ptr(), K_MLP_GEMM
            if l == L - 1 and spec.pool:
                gbuf_f = torch.empty(2, T, cout, device=dev, dtype=torch.float32)
                gbuf_i = torch.empty(2, T, cout, device=dev, dtype=torch.int32)
                g.gmax, g.gmin, g.amax, g.amin = gbuf_f[0].data_ptr(), gbuf_f[1].data_ptr(), gbuf_i[0].data_ptr(), gbuf_i[1].data_ptr()
            check(lib.papc_pg_gemm_f32(ctypes.byref(g), st), "papc_pg_gemm_f32")
            rm, rv = (bn_buffers[l] if bn_buffers is not None else (None, None))
            gamma, beta = params[4 * l + 2], params[4 * l + 3]
            if l < L - 1:
                # BN statistics of this layer folded in the prologue of the NEXT layer's operand prep (relu(bn(y)) -> planes)
                P = _planes(lib, M, cout, dev)
                if want_bwd:
                    PT[l + 1] = _planes(lib, cout, M, dev)
                prep(mode=BNRELU, M=M, C=cout, x=y.data_ptr(), ldx=cout, stats=stats.data_ptr(), parts=T, gamma=ptr(gamma), beta=ptr(beta),
                     eps=spec.eps, momentum=spec.momentum, running_mean=ptr(rm), running_var=ptr(rv), mean=cst[0].data_ptr(),
                     invstd=cst[1].data_ptr(), scale=cst[2].data_ptr(), shift=cst[3].data_ptr(), planes=P.data_ptr(), planes_t=ptr(PT[l + 1]))
            elif not spec.pool:
                # a point-wise stack (feature propagation, pointnet2_basic_layers.py:331-333): relu(bn(y)) of the last layer is the result
                check(lib.papc_bn_finalize_f32(stats.data_ptr(), T, M, cout, ptr(gamma), ptr(beta), spec.eps, spec.momentum, cst[0].data_ptr(),
                                               cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(), ptr(rm), ptr(rv), st), "papc_bn_finalize_f32")
                out = torch.empty(M, cout, device=dev, dtype=torch.float32)
                check(lib.papc_bn_relu_f32(y.data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(), M, cout, out.data_ptr(), st), "papc_bn_relu_f32")
                argmax = None
            else:
                out = torch.empty(T, cout, device=dev, dtype=torch.float32)
                argmax = torch.empty(T, cout, device=dev, dtype=torch.int32)
                check(lib.papc_pg_final_f32(stats.data_ptr(), T, M, cout, ptr(gamma), ptr(beta), spec.eps, spec.momentum, cst[0].data_ptr(),
                                            cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(), ptr(rm), ptr(rv), gbuf_f[0].data_ptr(),
                                            gbuf_f[1].data_ptr(), gbuf_i[0].data_ptr(), gbuf_i[1].data_ptr(), T, out.data_ptr(),
                                            argmax.data_ptr(), st), "papc_pg_final_f32")
            ys.append(y)
            consts.append(cst)
        ctx.spec, ctx.L, ctx.ch, ctx.plain, ctx.in_grad, ctx.n_in = spec, L, ch, plain, in_grad, n_in
        ctx.n_pt = sum(t is not None for t in PT)
        ctx.wt_mask = [t is not None for t in wtp]
        ysel = gbuf_f[0] if spec.pool else None              # raw y at the argmax (left there by papc_pg_final_f32)
        ctx.save_for_backward(argmax, ysel, *params, *ys, *consts, *[t for t in PT if t is not None], *[t for t in wtp if t is not None])
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        st = stream_ptr()
        spec, L, ch = ctx.spec, ctx.L, ctx.ch
        saved = ctx.saved_tensors
        argmax, ysel = saved[:2]
        params = saved[2:2 + 4 * L]
        ys = saved[2 + 4 * L: 2 + 5 * L]
        consts = saved[2 + 5 * L: 2 + 6 * L]
        o = 2 + 6 * L
        PT = list(saved[o:o + ctx.n_pt])
        assert len(PT) == L
        wts = iter(saved[o + ctx.n_pt:])
        wtp = [next(wts) if m else None for m in ctx.wt_mask]
        dev = gout.device
        M = spec.M
        T = M // GROUP
        nst = M // 32
        gout = gout.contiguous().float()
        grads = [None] * (4 * L)
        fold = []
        keep = []                                              # partial buffers stay alive until the fold is enqueued
        dz = red = None
        grad_in = None

        def prep(**kw):
            a = PgPrep()
            for k, v in kw.items():
                setattr(a, k, v)
            check(lib.papc_pg_prep_rows_f32(ctypes.byref(a), st), "papc_pg_prep_rows_f32")

        for l in range(L - 1, -1, -1):
            cout, cin = ch[l + 1], ch[l]
            cst = consts[l]
            tgt = spec.grad_targets[4 * l: 4 * l + 4] if spec.grad_targets is not None else None
            inplace = tgt is not None and all(t is not None for t in tgt)
            gb_inplace = tgt is not None and tgt[2] is not None and tgt[3] is not None
            if gb_inplace:
                dgamma_p, dbeta_p = tgt[2].data_ptr(), tgt[3].data_ptr()
            else:
                dgb = torch.empty(2, cout, device=dev, dtype=torch.float32)
                dgamma_p, dbeta_p = dgb[0].data_ptr(), dgb[1].data_ptr()
            need_dx = l > 0 or ctx.in_grad
            # dY of this layer as planes, both orientations; c1 / c2 / dgamma / dbeta folded in the prologue
            dyp = _planes(lib, M, cout, dev) if need_dx else None
            dypt = _planes(lib, cout, M, dev)
            common = dict(M=M, C=cout, x=ys[l].data_ptr(), ldx=cout, mean=cst[0].data_ptr(), invstd=cst[1].data_ptr(), scale=cst[2].data_ptr(),
                          shift=cst[3].data_ptr(), dgamma=dgamma_p, dbeta=dbeta_p, accumulate=int(gb_inplace), planes=ptr(dyp), planes_t=dypt.data_ptr())
            if l == L - 1 and not spec.pool:
                # dense upstream gradient: the layer's BN-backward sums in a pass of their own (T partial rows, folded in the prep's prologue)
                red = torch.empty(T, 2, cout, device=dev, dtype=torch.float32)
                check(lib.papc_bn_bwd_reduce_f32(DZ_DENSE, gout.data_ptr(), None, None, 1, ys[l].data_ptr(), cst[0].data_ptr(), cst[1].data_ptr(),
                                                 cst[2].data_ptr(), cst[3].data_ptr(), M, cout, T, red.data_ptr(), st), "papc_bn_bwd_reduce_f32")
                prep(mode=DY_DENSE, dz=gout.data_ptr(), red=red.data_ptr(), red_parts=T, **common)
            elif l == L - 1:
                prep(mode=DY_MAX, gout=gout.data_ptr(), ysel=ysel.data_ptr(), argmax=argmax.data_ptr(), K=GROUP, **common)
            else:
                prep(mode=DY_DENSE, dz=dz.data_ptr(), red=red.data_ptr(), red_parts=T, **common)
            # ---- dX
            if l > 0:
                pc = consts[l - 1]
                dz_prev = torch.empty(M, cin, device=dev, dtype=torch.float32)
                red_prev = torch.empty(T, 2, cin, device=dev, dtype=torch.float32)
                g = PgGemm()
                g.epi, g.a, g.b, g.R1, g.R2, g.K = EPI_RED, dyp.data_ptr(), wtp[l].data_ptr(), M, cin, cout
                g.c, g.ldc, g.split, g.split_stride, g.stats, g.family = dz_prev.data_ptr(), cin, 1, 0, red_prev.data_ptr(), K_BWD_DX
                g.y_prev, g.mean, g.invstd, g.scale, g.shift = ys[l - 1].data_ptr(), pc[0].data_ptr(), pc[1].data_ptr(), pc[2].data_ptr(), pc[3].data_ptr()
                check(lib.papc_pg_gemm_f32(ctypes.byref(g), st), "papc_pg_gemm_f32")
            elif ctx.in_grad:
                n_in = ctx.n_in
                grad_in = torch.empty(M, n_in, device=dev, dtype=torch.float32)
                g = PgGemm()
                g.epi, g.a, g.b, g.R1, g.R2, g.K = EPI_STORE, dyp.data_ptr(), wtp[0].data_ptr(), M, n_in, cout
                g.c, g.ldc, g.split, g.split_stride, g.family = grad_in.data_ptr(), n_in, 1, 0, K_BWD_DX
                check(lib.papc_pg_gemm_f32(ctypes.byref(g), st), "papc_pg_gemm_f32")
            # ---- dW = dY^T . input: contraction over the M rows, split over workgroups, partials folded at the end
            split = _split_for(cout, cin, nst)
            part = torch.empty(split, cout * cin, device=dev, dtype=torch.float32)
            keep.append(part)
            g = PgGemm()
            g.epi, g.a, g.b, g.R1, g.R2, g.K = EPI_STORE, dypt.data_ptr(), PT[l].data_ptr(), cout, cin, M
            g.c, g.ldc, g.split, g.split_stride, g.family = part.data_ptr(), cin, split, cout * cin, K_BWD_DW
            check(lib.papc_pg_gemm_f32(ctypes.byref(g), st), "papc_pg_gemm_f32")
            if inplace:
                fold.append((part.data_ptr(), split, cout * cin, cout * cin, tgt[0].data_ptr(), 1))
            else:
                dw = torch.empty(cout, cin, device=dev, dtype=torch.float32)
                fold.append((part.data_ptr(), split, cout * cin, cout * cin, dw.data_ptr(), 0))
                grads[4 * l + 0] = dw.reshape(params[4 * l].shape)
                # a bias feeding a train-mode BN has gradient exactly 0
                grads[4 * l + 1] = None if (tgt is not None and tgt[1] is not None) else _lib.zeros((cout,), dev)
                if not gb_inplace:
                    grads[4 * l + 2] = dgb[0]
                    grads[4 * l + 3] = dgb[1]
            if l > 0:
                dz, red = dz_prev, red_prev
        for j0 in range(0, len(fold), 8):
            chunk = fold[j0:j0 + 8]
            arr = (PgFoldJob * len(chunk))()
            for a, (pp, ns, stride, n, outp, acc) in zip(arr, chunk):
                a.partial, a.nsplit, a.stride, a.n, a.out, a.accumulate = pp, ns, stride, n, outp, acc
            check(lib.papc_pg_fold_f32(arr, len(chunk), st), "papc_pg_fold_f32")
        grad_feats = grad_x = None
        if grad_in is not None:
            if ctx.plain:
                grad_x = grad_in
            else:
                grad_feats = grad_in.view(spec.B, spec.N, spec.D)
        return (None, None, None, None, grad_feats, None, grad_x) + tuple(grads)
