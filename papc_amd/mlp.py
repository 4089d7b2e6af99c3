"""Shared pointwise MLP stack -- relu(bn(conv1x1(x))) x L, then max over the K rows of each group -- as ONE
autograd node driving the HIP kernels (include/papc_hip.h: papc_mlp_gemm_f32, papc_bn_finalize_f32,
papc_bn_relu_max_f32 and the papc_*bwd* family).

Reference: /root/reference/PAPC/models/layers/pointnet2_basic_layers.py:214-219 (SSG), :271-276 (MSG),
/root/reference/PAPC/models/classify/pointnet_base/pointnet_base.py:7-25,44 (PointNet-Basic Conv1D stack).

What is stored between forward and backward: only the pre-BN conv outputs y_l [M,C_l], the per-channel BN
constants and the argmax of the final max.  Grouped input rows, BN+ReLU activations and dY are recomputed
inside the operand loads of the MFMA kernels.
"""
import ctypes

import torch

from . import _lib
from ._lib import BwdDy, BwdRed, GroupMax, GroupSrc, ReduceJob, ScatterDst, check, ptr, stream_ptr

A_PLAIN, A_BNRELU, A_GROUP, A_XYZ = 0, 1, 2, 6
DZ_DENSE, DZ_MAX = 0, 1


class StackSpec:
    """Static description of one stack invocation (not a tensor; passed through autograd untouched)."""

    def __init__(self, B, N, S, K, D, xyz_first, eps=1e-5, momentum=0.9, cut_gather_grad=False, pool=True, eval_bn=False):
        self.B, self.N, self.S, self.K, self.D = B, N, S, K, D
        # pool=False: no max over K -- the stack returns relu(bn_L(.)) for every row (PointNetFeaturePropagation's
        # Conv1D stack, pointnet2_basic_layers.py:330-333)
        self.pool = bool(pool)
        # eval_bn=True: the BatchNorms normalise with their RUNNING statistics and leave them untouched (a registered norm of the
        # source under model.eval(): pointnet_base.py:8-24, pillars.py:24); needs bn_buffers.  The SA / FP layers never set it: the
        # source keeps their norms in plain lists, so .eval() never reaches them (pointnet2_basic_layers.py:185-191).
        self.eval_bn = bool(eval_bn)
        self.xyz_first = bool(xyz_first)
        self.eps, self.momentum = float(eps), float(momentum)
        self.cut_gather_grad = cut_gather_grad
        self.M = B * S * K
        # optional: per-parameter gradient buffers (contiguous, same shape as the parameter) that the backward
        # accumulates into directly instead of returning gradients to autograd (see grad_targets_of)
        self.grad_targets = None
        # optional: W^T operands precomputed for THIS forward pass by the caller (precompute_wt); None = transpose in the backward
        self.wt_table = None
        # optional: (xc, gram_partial) of xyz_pregroup for these very (xyz, new_xyz, idx): the coordinates-only first layer skips its grouping pass
        self.xyz_pre = None
        # optional: compact.CompactPlan of these very idx lists -- the stack then runs on the distinct neighbours only (csrc/compact.hip)
        self.compact = None
        self.plists = None             # optional compact.PointLists of (xyz, new_xyz, idx): the gather-add backward as a segmented sum (stack.py)


def _group_src(spec, xyz, new_xyz, feats, idx):
    g = GroupSrc()
    g.xyz = xyz.data_ptr()
    g.sb, g.sn, g.sc = xyz.stride(0), xyz.stride(1), xyz.stride(2)
    g.new_xyz = new_xyz.data_ptr()
    g.feats = ptr(feats)
    g.idx = ptr(idx)
    g.N, g.S, g.K, g.D = spec.N, spec.S, spec.K, spec.D
    g.xyz_first = int(spec.xyz_first)
    return g


import os
_FUSE_RED = os.environ.get("PAPC_NO_RED") != "1"            # A/B switch for the BN-backward reduce fused into the dX epilogue
_FUSE_GMAX = os.environ.get("PAPC_NO_GMAX") != "1"          # A/B switch for the fused neighbourhood-max epilogue
_RESIDENT_WGS = int(os.environ.get("PAPC_PARTS", "512"))   # persistent-grid size (tuning knob shared with the C side)
_LIN_GATHER = os.environ.get("PAPC_LIN_GATHER", "1") == "1"    # first grouped layer: linear map per source point, then gather-add (lingather.hip)
_NOSTORE = os.environ.get("PAPC_NOSTORE", "1") == "1"   # the max-pooled last layer without its [M, C] output where the library has all three flavours (papc_mlp_max_nostore_ok)
_SPARSE_MAX = os.environ.get("PAPC_SPARSE_MAX", "0") == "1"   # dX of the max-pooled last layer without reading its output y (papc_mlp_bwd_dx_max_f32)
_DW_WGS = int(os.environ.get("PAPC_DW_WGS", "512"))         # workgroups of one dW launch (row chunks x output tiles)
_PY_ORCH = os.environ.get("PAPC_PY_ORCH", "0") == "1"      # this module's own launch sequence instead of the library's papc_sa_mlp_fwd / _bwd (stack.py)
_XYZ1 = os.environ.get("PAPC_XYZ1", "1") == "1"             # coordinates-only first layer through its input moments, never materialised (xyz1.hip)
_XYZ_FUSE = os.environ.get("PAPC_XYZ_FUSE", "1") == "1"   # A/B switch (library orchestration only): 0 = the dX above a coordinates-only first layer is stored and read back by papc_xyz_l1_bwd_f32


def _dw_rows_per_chunk(M, cout, cin, min_rows=256):
    """Row-chunk size for the dW kernel: ONE residency wave of workgroups (256 CUs x 2 per CU = 512) in total, so
    no tail round; chunks of >= 256 rows -- >= 64 for the gather-add layer's dW_f over the B N source points (sa_mlp.hip::dw_rows_per_chunk:
    the same rule, bit for bit the same partial sums)."""
    wide = 128 < cin <= 160
    tiles = ((cout + 127) // 128) * (1 if wide else (cin + 127) // 128)
    want = max(1, _DW_WGS // tiles)
    rpc = (M + want - 1) // want
    rpc = max(min_rows, ((rpc + 63) // 64) * 64)
    return rpc


# ---- W^T operands of the dX GEMMs, precomputed for several stacks in one launch -------------------------------------------------
# A stack's backward needs the transposes of its layers' weights (one papc_transpose_batch_f32 per stack).  A model that runs several
# stacks per step can transpose ALL of them in one launch before its forward (precompute_wt) and hand the resulting table to exactly
# the stack invocations of THAT forward pass (StackSpec.wt_table, kept on the autograd node).  There is no module-level state: a stack
# called without a table (a layer used on its own, a later forward after the optimiser moved the weights) transposes locally, so a
# stale transpose can never be picked up.


def precompute_wt(weights, feat_blocks=()):
    """weights: conv / linear weights [Cout, Cin(, 1...)] whose [Cin, Cout] transposes the coming backward passes will need.
    ``feat_blocks``: (weight [Cout, D + 3(, 1...)], xyz_first) of stacks whose first layer may run as a gather-add: the feature block -- columns
    3.. of an SSG layer's weight (coordinates first), ..D of an MSG branch's -- is made contiguous in the same launch (key
    ``("f", weight.data_ptr())``, value ``(block, xyz_first)``): the copy the stack's forward otherwise launches itself.
    Returns the table {weight.data_ptr(): W^T tensor, ...} to pass on as ``StackSpec.wt_table`` for this forward pass only."""
    lib = _lib.load()
    st = stream_ptr()
    table = {}
    ok = lambda w: w.is_cuda and w.is_contiguous() and w.dtype == torch.float32      # noqa: E731
    jobs = []        # (src pointer, source row stride, destination, rows, cols, copy)
    for w in weights:
        if ok(w):
            cout = w.shape[0]
            cin = w.numel() // cout
            t = torch.empty(cin, cout, device=w.device, dtype=torch.float32)
            table[w.data_ptr()] = t
            jobs.append((w.data_ptr(), cin, t, cout, cin, 0))
    for w, xyz_first in feat_blocks:
        cout = w.shape[0]
        cin = w.numel() // cout
        if ok(w) and cin > 3:
            t = torch.empty(cout, cin - 3, device=w.device, dtype=torch.float32)
            table[("f", w.data_ptr())] = (t, bool(xyz_first))
            jobs.append((w.data_ptr() + (12 if xyz_first else 0), cin, t, cout, cin - 3, 1))
    for g0 in range(0, len(jobs), 8):
        grp = jobs[g0:g0 + 8]
        n = len(grp)
        srcs, dsts = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        lds, rws, cls, cps = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
        for i, (src, ld, t, rows, cols, cp) in enumerate(grp):
            srcs[i], lds[i], dsts[i], rws[i], cls[i], cps[i] = src, ld, t.data_ptr(), rows, cols, cp
        check(lib.papc_transpose_batch_ld_f32(srcs, lds, dsts, rws, cls, cps, n, st), "papc_transpose_batch_ld_f32")
    return table


def xyz_first_layer_ok(M, c1, c2, n_layers):
    """whether a coordinates-only stack of these widths takes the input-moment path of its first layer (xyz1.hip)"""
    return bool(_XYZ1 and n_layers >= 3 and _lib.load().papc_mlp_xyz_ok(M, c1, c2))


def xyz_pregroup(xyz, new_xyz, idx, out=None):
    """The weight-independent part of that path -- grouped centred coordinates xc [M, 4] and their float64 partial moments -- so a
    training loop can compute it with the sampling pyramid (layers.PointNetSetAbstraction.sample).  xyz [B,N,3] strided, new_xyz [B,S,3],
    idx [B,S,K] int32.  ``out`` = optional preallocated (xc [M,4], gram [1,16] float64)."""
    lib = _lib.load()
    B, N = xyz.shape[0], xyz.shape[1]
    S, K = idx.shape[1], idx.shape[2]
    M = B * S * K
    dev = xyz.device
    if out is None:
        out = (torch.empty(M, 4, device=dev, dtype=torch.float32), torch.empty(1, 16, device=dev, dtype=torch.float64))
    gpart = torch.empty(lib.papc_xyz_parts(M), 16, device=dev, dtype=torch.float64)
    g = GroupSrc()
    g.xyz = xyz.data_ptr()
    g.sb, g.sn, g.sc = xyz.stride(0), xyz.stride(1), xyz.stride(2)
    g.new_xyz, g.feats, g.idx = new_xyz.data_ptr(), 0, idx.data_ptr()
    g.N, g.S, g.K, g.D, g.xyz_first = N, S, K, 0, 1
    check(lib.papc_xyz_group_f32(ctypes.byref(g), B, ptr(out[0]), ptr(gpart), stream_ptr()), "papc_xyz_group_f32")
    check(lib.papc_xyz_gram_fold_f32(ptr(gpart), gpart.shape[0], ptr(out[1]), stream_ptr()), "papc_xyz_gram_fold_f32")
    return out


class SharedMLPMax(torch.autograd.Function):
    """out[g, :] = max_{k<K} relu(bn_L(conv_L(... relu(bn_1(conv_1(rows)))...)))   with rows gathered on the fly.

    apply(spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, w1, b1, gamma1, beta1, w2, ...)
      x_rows   None, or plain input rows [M, Cin] (then xyz/new_xyz/feats/idx are ignored: PFNLayer)
      xyz      [B,N,3] any strides        new_xyz [B,S,3] contiguous
      feats    [B,N,D] contiguous or None idx     [B,S,K] int32 or None (identity: S=1, K=N)
      w_l      [C_l, C_{l-1}] (conv weight, trailing 1x1 dims dropped); b_l, gamma_l, beta_l [C_l]
      bn_buffers: list of (running_mean, running_var) per layer (updated in place) or None
    returns [B*S, C_L]   (spec.pool=False: [M, C_L], every row's activation)
    """

    @staticmethod
    def forward(ctx, spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, *params):
        lib = _lib.load()
        st = stream_ptr()
        dev = params[0].device
        L = len(params) // 4
        M = spec.M
        parts = lib.papc_mlp_gemm_parts(M)
        plain = x_rows is not None
        grp = None if plain else _group_src(spec, xyz, new_xyz, feats, idx)
        ys, consts = [], []
        prev_y, prev_sc, prev_sh = None, None, None
        cin = x_rows.shape[1] if plain else spec.D + 3
        cin0 = cin
        ev = spec.eval_bn
        assert not ev or bn_buffers is not None, "eval_bn needs the running statistics"
        lin0 = (_LIN_GATHER and not ev and not plain and idx is not None and feats is not None and L >= 2 and spec.D % 4 == 0 and spec.D >= 16
                and params[0].shape[0] % 4 == 0 and params[0].shape[0] <= 256 and feats.is_contiguous())
        # coordinates-only first layer (D = 0): its [M, C1] output is never stored -- BN statistics from the inputs' second moments, the
        # layer folded into the second layer's operand (xyz1.hip)
        xyz1 = (not ev and not plain and idx is not None and feats is None and spec.D == 0
                and xyz_first_layer_ok(M, params[0].shape[0], params[4].shape[0] if L >= 2 else 0, L))
        xc = wf = gram = None
        # compacted stack: distinct neighbours only, a weight on each group's first row (compact.py); the row count is a device-side number
        cp = spec.compact if (lin0 and spec.pool and spec.compact is not None) else None
        if cp is not None:
            from . import compact as _cpm
            if not _cpm.stack_ok(M, spec.K, [params[4 * l].shape[0] for l in range(L)]):
                cp = None
        R_c = lib.papc_compact_corr_parts() if cp is not None else 0     # extra statistics rows: the copies' share
        if cp is not None:
            grp.cidx, grp.seg_grp, grp.rows_dev = cp.cidx.data_ptr(), cp.seg_grp.data_ptr(), cp.rows.data_ptr()
        for l in range(L):
            w, b, gamma, beta = params[4 * l: 4 * l + 4]
            cout = w.shape[0]
            w2 = w.reshape(cout, cin)
            assert w2.is_contiguous()
            stats = None if ev else torch.empty(parts + R_c, 2, cout, device=dev, dtype=torch.float32)
            parts_l = parts
            gm_ref = None
            if cp is None and l == L - 1 and spec.pool and _FUSE_GMAX and not ev and lib.papc_mlp_gemm_gmax_ok(M, cout, spec.K):
                # last layer: the neighbourhood max is reduced in the GEMM epilogue (per-group max/min of the raw output)
                G_ = M // spec.K
                gbuf_f = torch.empty(2, G_, cout, device=dev, dtype=torch.float32)
                gbuf_i = torch.empty(2, G_, cout, device=dev, dtype=torch.int32)
                gm = GroupMax()
                gm.gmax, gm.gmin = gbuf_f[0].data_ptr(), gbuf_f[1].data_ptr()
                gm.amax, gm.amin = gbuf_i[0].data_ptr(), gbuf_i[1].data_ptr()
                gm.K = spec.K
                gm_ref = ctypes.byref(gm)
            if (gm_ref is not None and _NOSTORE and L >= 2 and l == L - 1 and prev_y is not None and not (l == 1 and xyz1)
                    and lib.papc_mlp_max_nostore_ok(M, cin, cout, spec.K)):
                y = None                           # the pooled layer's own output is never stored: extrema + statistics forward, input sums backward
            else:
                y = None if (l == 0 and xyz1) else torch.empty(M, cout, device=dev, dtype=torch.float32)
            if l == 0 and xyz1:
                y = stats = None
                nparts = 1                         # (the moments arrive folded: xyz_pregroup)
                if spec.xyz_pre is not None:       # grouped with the sampling pyramid (possibly on another stream, one step ahead)
                    xc, gpart = spec.xyz_pre
                    assert tuple(xc.shape) == (M, 4) and tuple(gpart.shape) == (1, 16)
                else:
                    xc, gpart = xyz_pregroup(xyz, new_xyz, idx)
                cst = torch.empty(4, cout, device=dev, dtype=torch.float32)
                wf = torch.empty(cout, 4, device=dev, dtype=torch.float32)
                gram = torch.empty(16, device=dev, dtype=torch.float64)
                rm, rv = (bn_buffers[l] if bn_buffers is not None else (None, None))
                check(lib.papc_xyz_l1_finalize_f32(ptr(gpart), nparts, M, ptr(w2), cin, 0, ptr(b), ptr(gamma), ptr(beta), spec.eps, spec.momentum,
                                                   cout, cst[0].data_ptr(), cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(), ptr(rm), ptr(rv),
                                                   ptr(wf), ptr(gram), st), "papc_xyz_l1_finalize_f32")
                ys.append(None)
                consts.append(cst)
                prev_y, prev_sc, prev_sh = None, cst[2], cst[3]
                cin = cout
                continue
            if l == 1 and xyz1:
                check(lib.papc_mlp_gemm_f32(A_XYZ, ptr(xc), 4, None, ptr(wf), None, ptr(w2), ptr(b), M, cin, cout,
                                            ptr(y), ptr(stats), gm_ref, st), "papc_mlp_gemm_f32")
            elif l == 0 and plain:
                check(lib.papc_mlp_gemm_f32(A_PLAIN, ptr(x_rows), cin, None, None, None, ptr(w2), ptr(b), M, cin, cout,
                                            ptr(y), ptr(stats), gm_ref, st), "papc_mlp_gemm_f32")
            elif l == 0 and lin0:
                # W_f feats_j depends on the source point only: one [B*N, D] x [D, cout] product (the row GEMM kernel, plain input, no
                # bias / statistics) on the feature block of the weight, then a streaming gather-add
                fcol0 = 3 if spec.xyz_first else 0
                wf = torch.empty(cout, spec.D, device=dev, dtype=torch.float32)
                check(lib.papc_copy2d_f32(w2.data_ptr() + 4 * fcol0, cin, ptr(wf), spec.D, cout, spec.D, 0, st), "papc_copy2d_f32")
                BN_ = spec.B * spec.N
                P = torch.empty(BN_, cout, device=dev, dtype=torch.float32)
                check(lib.papc_mlp_gemm_f32(A_PLAIN, ptr(feats), spec.D, None, None, None, ptr(wf), None, BN_, spec.D, cout, ptr(P), None,
                                            None, st), "papc_mlp_gemm_f32")
                parts_l = lib.papc_lingather_parts(M)
                stats = torch.empty(parts_l + R_c, 2, cout, device=dev, dtype=torch.float32)
                check(lib.papc_lingather_fwd_f32(ptr(P), ctypes.byref(grp), spec.B, ptr(w2), cin, 0 if spec.xyz_first else spec.D,
                                                 ptr(b), cout, ptr(y), ptr(stats), st), "papc_lingather_fwd_f32")
            elif l == 0:
                check(lib.papc_mlp_gemm_f32(A_GROUP, None, 0, ctypes.byref(grp), None, None, ptr(w2), ptr(b), M, cin, cout,
                                            ptr(y), ptr(stats), gm_ref, st), "papc_mlp_gemm_f32")
            elif cp is not None:
                check(lib.papc_mlp_gemm_rows_f32(A_BNRELU, ptr(prev_y), cin, None, ptr(prev_sc), ptr(prev_sh), ptr(w2), ptr(b), M, cin,
                                                 cout, ptr(y), ptr(stats), None, cp.rows.data_ptr(), st), "papc_mlp_gemm_rows_f32")
            else:
                check(lib.papc_mlp_gemm_f32(A_BNRELU, ptr(prev_y), cin, None, ptr(prev_sc), ptr(prev_sh), ptr(w2), ptr(b), M, cin,
                                            cout, ptr(y), ptr(stats), gm_ref, st), "papc_mlp_gemm_f32")
            if cp is not None:      # what the copies add to this layer's statistics: R_c extra partial rows behind the kernel's own
                check(lib.papc_bn_stats_corr_f32(ptr(y), cout, cp.start.data_ptr(), cp.coef.data_ptr(), cp.G,
                                                 stats.data_ptr() + 4 * parts_l * 2 * cout, st), "papc_bn_stats_corr_f32")
                parts_l += R_c
            cst = torch.empty(4, cout, device=dev, dtype=torch.float32)  # mean, invstd, scale, shift
            rm, rv = (bn_buffers[l] if bn_buffers is not None else (None, None))
            if ev:
                check(lib.papc_bn_eval_consts_f32(ptr(rm), ptr(rv), ptr(gamma), ptr(beta), spec.eps, cout, cst[0].data_ptr(),
                                                  cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(), st), "papc_bn_eval_consts_f32")
            else:
                check(lib.papc_bn_finalize_f32(ptr(stats), parts_l, M, cout, ptr(gamma), ptr(beta), spec.eps, spec.momentum,
                                               cst[0].data_ptr(), cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr(),
                                               ptr(rm), ptr(rv), st), "papc_bn_finalize_f32")
            ys.append(y)
            consts.append(cst)
            prev_y, prev_sc, prev_sh = y, cst[2], cst[3]
            cin = cout
        G = spec.B * spec.S
        argmax = None
        if not spec.pool:
            out = torch.empty(M, cin, device=dev, dtype=torch.float32)
            check(lib.papc_bn_relu_f32(ptr(prev_y), ptr(prev_sc), ptr(prev_sh), M, cin, ptr(out), st), "papc_bn_relu_f32")
        else:
            out = torch.empty(G, cin, device=dev, dtype=torch.float32)
            argmax = torch.empty(G, cin, device=dev, dtype=torch.int32)
        ysel_c = None
        if not spec.pool:
            pass
        elif cp is not None:            # ragged groups: the max runs over each group's physical rows; argmax = absolute row
            ysel_c = torch.empty(G, cin, device=dev, dtype=torch.float32)
            check(lib.papc_bn_relu_max_seg_f32(ptr(prev_y), cin, cp.start.data_ptr(), ptr(prev_sc), ptr(prev_sh), G, ptr(out), ptr(argmax),
                                               ptr(ysel_c), st), "papc_bn_relu_max_seg_f32")
        elif gm_ref is not None:
            check(lib.papc_bn_select_max_f32(gbuf_f[0].data_ptr(), gbuf_f[1].data_ptr(), gbuf_i[0].data_ptr(), gbuf_i[1].data_ptr(),
                                             ptr(prev_sc), ptr(prev_sh), G, cin, ptr(out), ptr(argmax), st), "papc_bn_select_max_f32")
        else:
            check(lib.papc_bn_relu_max_f32(ptr(prev_y), ptr(prev_sc), ptr(prev_sh), G, spec.K, cin, ptr(out), ptr(argmax), st),
                  "papc_bn_relu_max_f32")
        ctx.spec = spec
        ctx.L = L
        ctx.wt_table = spec.wt_table
        ctx.feats_needs_grad = feats is not None and feats.requires_grad and not spec.cut_gather_grad
        ctx.x_needs_grad = plain and x_rows.requires_grad
        ctx.cin0 = cin0
        ctx.lin0 = lin0
        ctx.xyz1 = xyz1
        ctx.nostore = spec.pool and L >= 1 and ys[L - 1] is None and not (xyz1 and L == 1)
        ysel = gbuf_f[0] if (spec.pool and gm_ref is not None) else ysel_c   # raw y at the argmax (left in gmax by select_max)
        ctx.compact = cp
        if xyz1:
            ys[0] = xc                 # (slot of the first layer's output, which does not exist: the grouped coordinates instead)
            consts = consts + [wf, gram]
        ctx.save_for_backward(xyz, new_xyz, feats, idx, x_rows, argmax, ysel, *params, *ys, *consts)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        st = stream_ptr()
        spec, L = ctx.spec, ctx.L
        saved = ctx.saved_tensors
        xyz, new_xyz, feats, idx, x_rows, argmax, ysel = saved[:7]
        params = saved[7:7 + 4 * L]
        ys = saved[7 + 4 * L: 7 + 5 * L]
        consts = saved[7 + 5 * L: 7 + 6 * L]
        xyz1 = ctx.xyz1
        xc = wf = gram = None
        if xyz1:
            xc, (wf, gram) = ys[0], saved[7 + 6 * L: 7 + 6 * L + 2]
        plain = x_rows is not None
        dev = gout.device
        M = spec.M
        gout = gout.contiguous().float()
        grp = None if plain else _group_src(spec, xyz, new_xyz, feats, idx)
        cp = ctx.compact
        if cp is not None:
            grp.cidx, grp.seg_grp, grp.rows_dev = cp.cidx.data_ptr(), cp.seg_grp.data_ptr(), cp.rows.data_ptr()
        n_parts = min(_RESIDENT_WGS, (M + 127) // 128)
        grads = [None] * (4 * L)
        reduce_jobs = []   # (partial tensor, n_chunks, ld, n1, out1 ptr, n2, out2 ptr, accumulate): see the dW section
        grad_feats = None
        grad_x = None
        dz = None
        fused_red = None
        gemm_parts = lib.papc_mlp_gemm_parts(M)
        # W^T operands of every dX GEMM of the stack: one batched transpose launch
        need_wt = [l for l in range(L) if l > 0 or (plain and ctx.x_needs_grad) or ((not plain) and ctx.feats_needs_grad)]
        # last layer under the max: its dX can be formed from the [G, C] max-backward arrays and the layer's INPUT, without its output
        cL, cLi = params[4 * (L - 1)].shape[0], (params[4 * (L - 2)].shape[0] if L > 1 else 0)
        sparse_max = ctx.nostore or (_SPARSE_MAX and spec.pool and L > 1 and ysel is not None and M >= 32768 and cL % 16 == 0 and cLi % 4 == 0
                                     and cL + cLi <= 512)
        max_prep = None     # (psel, wcat, hbias, e) of the sparse-max backward, shared by its dW and dX
        if sparse_max:
            need_wt.remove(L - 1)
        if ctx.lin0 and 0 in need_wt:
            need_wt.remove(0)
        wts = {}
        if ctx.wt_table is not None:       # transposes handed over for this very forward pass (precompute_wt)
            for l in list(need_wt):
                t = ctx.wt_table.get(params[4 * l].data_ptr())
                cin_l = ctx.cin0 if l == 0 else params[4 * (l - 1)].shape[0]
                if t is not None and tuple(t.shape) == (cin_l, params[4 * l].shape[0]):
                    wts[l] = t
                    need_wt.remove(l)
        for g0 in range(0, len(need_wt), 8):
            grp_l = need_wt[g0:g0 + 8]
            n = len(grp_l)
            srcs, dsts, rws, cls = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
            for i, l in enumerate(grp_l):
                w = params[4 * l]
                cout = w.shape[0]
                cin = ctx.cin0 if l == 0 else params[4 * (l - 1)].shape[0]
                assert w.is_contiguous()
                wts[l] = torch.empty(cin, cout, device=dev, dtype=torch.float32)
                srcs[i], dsts[i], rws[i], cls[i] = w.data_ptr(), wts[l].data_ptr(), cout, cin
            check(lib.papc_transpose_batch_f32(srcs, dsts, rws, cls, n, st), "papc_transpose_batch_f32")
        for l in range(L - 1, -1, -1):
            w = params[4 * l]
            cout = w.shape[0]
            cin = ctx.cin0 if l == 0 else params[4 * (l - 1)].shape[0]
            cst = consts[l]
            c12 = torch.empty(2, cout, device=dev, dtype=torch.float32)
            tgt = spec.grad_targets[4 * l: 4 * l + 4] if spec.grad_targets is not None else None
            inplace = tgt is not None and all(t is not None for t in tgt)
            # the norm's two vectors can go in place on their own (a layer whose conv weight is a padded view keeps them off autograd)
            gb_inplace = tgt is not None and tgt[2] is not None and tgt[3] is not None and not spec.eval_bn
            if gb_inplace:   # accumulate straight into the parameters' .grad (flat-bucket views): no autograd add kernels
                dgamma_p, dbeta_p = tgt[2].data_ptr(), tgt[3].data_ptr()
            else:
                dgb = torch.empty(2, cout, device=dev, dtype=torch.float32)  # dgamma, dbeta
                dgamma_p, dbeta_p = dgb[0].data_ptr(), dgb[1].data_ptr()
            dy = BwdDy()
            if cp is not None:      # multiplicity weights, ragged groups, device-side row count
                dy.wrow, dy.seg_grp, dy.rows_dev = cp.wrow.data_ptr(), cp.seg_grp.data_ptr(), cp.rows.data_ptr()
            if l == L - 1 and not spec.pool:
                dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = DZ_DENSE, gout.data_ptr(), None, None, 1
            elif l == L - 1:
                dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = DZ_MAX, None, gout.data_ptr(), argmax.data_ptr(), spec.K
            else:
                dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = DZ_DENSE, dz.data_ptr(), None, None, 1
            if l == 0 and xyz1:
                # the whole backward of the coordinates-only layer from one pass over dz and the inputs' moments: p = dz [a > 0] summed
                # against the coordinates, then closed forms for dgamma / dbeta / dW (no y, no BN-backward sums from the layer above)
                nbp = lib.papc_xyz_bwd_parts(M)
                bpart = torch.empty(nbp, cout, 4, device=dev, dtype=torch.float32)
                check(lib.papc_xyz_l1_bwd_f32(ptr(dz), ptr(xc), ptr(wf), M, cout, ptr(bpart), st), "papc_xyz_l1_bwd_f32")
                if inplace:
                    dw0, acc0 = tgt[0].view(cout, cin), 1
                else:
                    dw0, acc0 = torch.empty(cout, cin, device=dev, dtype=torch.float32), 0
                if gb_inplace != bool(acc0):       # (one accumulate flag for the three outputs: fall back to fresh gamma / beta buffers)
                    dgb = torch.empty(2, cout, device=dev, dtype=torch.float32)
                    dgamma_p, dbeta_p = dgb[0].data_ptr(), dgb[1].data_ptr()
                    gb_fresh = True
                else:
                    gb_fresh = not gb_inplace
                check(lib.papc_xyz_l1_bwd_finalize_f32(ptr(bpart), nbp, M, cout, ptr(gram), ptr(w), cin, 0, ptr(params[1]), cst[0].data_ptr(),
                                                       cst[1].data_ptr(), cst[2].data_ptr(), dgamma_p, dbeta_p, ptr(dw0), acc0, st),
                      "papc_xyz_l1_bwd_finalize_f32")
                if not inplace:
                    grads[0] = dw0.reshape(w.shape)
                    grads[1] = None if (tgt is not None and tgt[1] is not None) else _lib.zeros((cout,), dev)
                if gb_fresh:
                    grads[2], grads[3] = dgb[0], dgb[1]
                break
            dy.y = ptr(ys[l])
            dy.mean, dy.invstd, dy.scale, dy.shift = (cst[i].data_ptr() for i in range(4))
            dy.c1, dy.c2 = c12[0].data_ptr(), c12[1].data_ptr()
            if fused_red is None:   # (sum p, sum p*xhat): separate pass, unless the dX kernel of layer l+1 already produced it
                red, red_parts = torch.empty(n_parts, 2, cout, device=dev, dtype=torch.float32), n_parts
                red_dz = ptr(ysel) if dy.dz_mode == DZ_MAX else dy.dz
                if sparse_max and l == L - 1:     # the same pass also writes the sparse operand of this layer's dX / dW
                    psel = torch.empty(M // spec.K, cout, device=dev, dtype=torch.float32)
                    check(lib.papc_bn_bwd_reduce_max_f32(ptr(ysel), dy.gout, dy.K, dy.mean, dy.invstd, dy.scale, dy.shift, M, cout, n_parts,
                                                         ptr(red), ptr(psel), st), "papc_bn_bwd_reduce_max_f32")
                else:
                    check(lib.papc_bn_bwd_reduce_f32(dy.dz_mode, red_dz, dy.gout, dy.argmax, dy.K, dy.y, dy.mean, dy.invstd, dy.scale,
                                                     dy.shift, M, cout, n_parts, ptr(red), st), "papc_bn_bwd_reduce_f32")
            else:
                red, red_parts = fused_red, gemm_parts
            check(lib.papc_bn_bwd_finalize_f32(ptr(red), red_parts, M, cout, dgamma_p, dbeta_p,
                                               c12[0].data_ptr(), c12[1].data_ptr(), int(gb_inplace) | (2 if spec.eval_bn else 0), st),
                  "papc_bn_bwd_finalize_f32")
            if l == 0 and ctx.lin0:
                # G[j] = sum of the dY rows that gathered point j (+ the xyz columns of dW, streamed); the D-wide products run on B*N rows
                BN_ = spec.B * spec.N
                parts_l = lib.papc_lingather_parts(M)
                dwx_part = torch.empty(parts_l, cout * 3, device=dev, dtype=torch.float32)
                Gs = _lib.zeros((BN_, cout), dev)
                check(lib.papc_lingather_bwd_f32(ctypes.byref(dy), ctypes.byref(grp), spec.B, cout, ptr(Gs), ptr(dwx_part), st),
                      "papc_lingather_bwd_f32")
                # the gradient lands in ONE [cout, cin] tensor -- the parameter's .grad view (accumulate) or a fresh one: the
                # coordinate block and the feature block are summed straight into their column ranges (no concat, no add)
                fcol0, xcol0 = (3, 0) if spec.xyz_first else (0, spec.D)
                if inplace:
                    dw, acc = tgt[0].view(cout, cin), 1
                else:
                    dw, acc = torch.empty(cout, cin, device=dev, dtype=torch.float32), 0
                check(lib.papc_reduce_partials_strided_f32(ptr(dwx_part), parts_l, cout * 3, cout, 3, dw.data_ptr() + 4 * xcol0, cin, acc, st),
                      "papc_reduce_partials_strided_f32")
                # dW_f = G^T feats on the library's own dW kernel: G plays dY with BN constants that make dY = dz (scale 1, shift huge
                # -> ReLU mask always on, c1 = c2 = 0)
                one, zero, big = _lib.const_vec(1.0, cout, dev), _lib.const_vec(0.0, cout, dev), _lib.const_vec(1e30, cout, dev)
                dyg = BwdDy()
                dyg.dz_mode, dyg.dz, dyg.gout, dyg.argmax, dyg.K = DZ_DENSE, Gs.data_ptr(), None, None, 1
                dyg.y = Gs.data_ptr()
                dyg.mean, dyg.invstd, dyg.scale, dyg.shift = zero.data_ptr(), one.data_ptr(), one.data_ptr(), big.data_ptr()
                dyg.c1, dyg.c2 = zero.data_ptr(), zero.data_ptr()
                rpc_g = _dw_rows_per_chunk(BN_, cout, spec.D, 64)
                n_chunks_g = (BN_ + rpc_g - 1) // rpc_g
                pld_g = cout * spec.D + cout
                part_g = torch.empty(n_chunks_g, pld_g, device=dev, dtype=torch.float32)
                check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dyg), A_PLAIN, ptr(feats), spec.D, None, None, None, BN_, spec.D, cout, rpc_g,
                                              part_g.data_ptr(), part_g.data_ptr() + 4 * cout * spec.D, pld_g, st), "papc_mlp_bwd_dw_f32")
                check(lib.papc_reduce_partials_strided_f32(ptr(part_g), n_chunks_g, pld_g, cout, spec.D, dw.data_ptr() + 4 * fcol0, cin, acc,
                                                           st), "papc_reduce_partials_strided_f32")
                if not inplace:           # (db: a bias feeding a train-mode BN has gradient exactly 0)
                    grads[0] = dw.reshape(w.shape)
                    grads[1] = None if (tgt is not None and tgt[1] is not None) else _lib.zeros((cout,), dev)   # (exact 0: nothing to add in place)
                    if not gb_inplace:
                        grads[2] = dgb[0]
                        grads[3] = dgb[1]
                if ctx.feats_needs_grad:
                    # grad_feats = G W_f: the same row GEMM with the transposed feature block as its weight
                    wt_full = ctx.wt_table.get(w.data_ptr()) if ctx.wt_table is not None else None
                    if wt_full is not None and tuple(wt_full.shape) == (cin, cout):
                        wft = wt_full[fcol0:fcol0 + spec.D]     # rows of the precomputed W^T [cin, cout]: the feature block, contiguous
                    else:
                        wft = torch.empty(spec.D, cout, device=dev, dtype=torch.float32)
                        check(lib.papc_copy2d_f32(w.data_ptr() + 4 * fcol0, cin, ptr(wft), cout, cout, spec.D, 1, st), "papc_copy2d_f32")
                    grad_feats = torch.empty(spec.B, spec.N, spec.D, device=dev, dtype=torch.float32)
                    check(lib.papc_mlp_gemm_f32(A_PLAIN, ptr(Gs), cout, None, None, None, ptr(wft), None, BN_, cout, spec.D, ptr(grad_feats),
                                                None, None, st), "papc_mlp_gemm_f32")
                break
            if sparse_max and l == L - 1:
                G = M // spec.K
                wcat = torch.empty(cin, cout + cin, device=dev, dtype=torch.float32)
                hb = torch.empty(cin, device=dev, dtype=torch.float32)
                eq = torch.empty(2, cout, device=dev, dtype=torch.float32)
                check(lib.papc_bn_max_prep_f32(None, None, cst[2].data_ptr(), cst[3].data_ptr(), cst[0].data_ptr(),
                                               cst[1].data_ptr(), c12[0].data_ptr(), c12[1].data_ptr(), ptr(w), ptr(params[4 * l + 1]),
                                               G, cout, cin, None, ptr(wcat), ptr(hb), eq[0].data_ptr(), eq[1].data_ptr(), st),
                      "papc_bn_max_prep_f32")      # (psel: written by the reduction above)
                max_prep = (psel, wcat, hb, eq)
            # ---- dW, db
            if ctx.nostore and l == L - 1:
                # no stored output: T = P'^T A, the Gram matrix of the input rows and their column sums in one pass, closed form for dW
                pc = consts[l - 1]
                ws = torch.empty(lib.papc_mlp_bwd_dw_max_ws_floats(M, cin, cout), device=dev, dtype=torch.float32)
                if inplace:
                    dw, acc = tgt[0].view(cout, cin), 1
                else:
                    dw, acc = torch.empty(cout, cin, device=dev, dtype=torch.float32), 0
                check(lib.papc_mlp_bwd_dw_max_f32(ptr(max_prep[0]), ptr(argmax), spec.K, ys[l - 1].data_ptr(), pc[2].data_ptr(), pc[3].data_ptr(),
                                                  ptr(w), max_prep[3][0].data_ptr(), cst[2].data_ptr(), c12[0].data_ptr(), M, cin, cout,
                                                  ptr(ws), ptr(dw), acc, st), "papc_mlp_bwd_dw_max_f32")
                if not inplace:
                    grads[4 * l + 0] = dw.reshape(w.shape)
                    grads[4 * l + 1] = None if (tgt is not None and tgt[1] is not None) else _lib.zeros((cout,), dev)   # (exact 0 under a train-mode BN)
                    if not gb_inplace:
                        grads[4 * l + 2] = dgb[0]
                        grads[4 * l + 3] = dgb[1]
            x1 = xyz1 and l == 1             # the input of this layer is the recomputed activation of the coordinates-only first layer
            if not (ctx.nostore and l == L - 1):
                rpc = 0
                if l > 0:                        # (the kernel's own preference where it has one: papc_mlp_bwd_dw_chunk_hint)
                    rpc = lib.papc_mlp_bwd_dw_chunk_hint(M, cin, cout, A_XYZ if x1 else A_BNRELU, dy.dz_mode, spec.K if dy.dz_mode == DZ_MAX else 0)
                elif not plain:
                    rpc = lib.papc_mlp_bwd_dw_chunk_hint(M, cin, cout, A_GROUP, dy.dz_mode, spec.K if dy.dz_mode == DZ_MAX else 0)
                if rpc <= 0:
                    rpc = _dw_rows_per_chunk(M, cout, cin)
                n_chunks = (M + rpc - 1) // rpc
                pld = cout * cin + cout          # one partial buffer: chunk rows are [dW (cout*cin) | db (cout)]
                part = torch.empty(n_chunks, pld, device=dev, dtype=torch.float32)
                dwp_p, dbp_p = part.data_ptr(), part.data_ptr() + 4 * cout * cin
                if l == 0 and plain:
                    check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dy), A_PLAIN, ptr(x_rows), cin, None, None, None, M, cin, cout, rpc,
                                                  dwp_p, dbp_p, pld, st), "papc_mlp_bwd_dw_f32")
                elif l == 0:
                    check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dy), A_GROUP, None, 0, ctypes.byref(grp), None, None, M, cin, cout, rpc,
                                                  dwp_p, dbp_p, pld, st), "papc_mlp_bwd_dw_f32")
                elif x1:
                    check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dy), A_XYZ, ptr(xc), 4, None, ptr(wf), None, M, cin, cout, rpc, dwp_p, dbp_p, pld, st),
                          "papc_mlp_bwd_dw_f32")
                else:
                    pc = consts[l - 1]
                    check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dy), A_BNRELU, ys[l - 1].data_ptr(), cin, None, pc[2].data_ptr(),
                                                  pc[3].data_ptr(), M, cin, cout, rpc, dwp_p, dbp_p, pld, st), "papc_mlp_bwd_dw_f32")
                # the partials of all layers are folded in ONE launch once the stack's last dW kernel is enqueued (papc_reduce_partials_batch_f32)
                if inplace:
                    reduce_jobs.append((part, n_chunks, pld, cout * cin, tgt[0].data_ptr(), cout, tgt[1].data_ptr(), 1))
                else:
                    dw = torch.empty(cout, cin, device=dev, dtype=torch.float32)
                    db = torch.empty(cout, device=dev, dtype=torch.float32)
                    if spec.eval_bn:      # no batch-mean term removes the bias direction: db = sum_m dy = scale * sum_m p (tiny [C] op)
                        check(lib.papc_reduce_partials2_f32(ptr(part), n_chunks, pld, cout * cin, ptr(dw), cout, ptr(db), 0, st),
                              "papc_reduce_partials2_f32")
                        db = cst[2] * dgb[1]
                    else:
                        reduce_jobs.append((part, n_chunks, pld, cout * cin, dw.data_ptr(), cout, db.data_ptr(), 0))
                    grads[4 * l + 0] = dw.reshape(w.shape)
                    grads[4 * l + 1] = None if (tgt is not None and tgt[1] is not None and not spec.eval_bn) else db
                    if not gb_inplace:
                        grads[4 * l + 2] = dgb[0]
                        grads[4 * l + 3] = dgb[1]
            # ---- dX
            fused_red = None
            if l > 0:
                wt = wts.get(l)
                dz_prev = torch.empty(M, cin, device=dev, dtype=torch.float32)
                # the dX kernel also accumulates layer l-1's BN-backward reductions over the dz it produces
                nr_ref = None
                if _FUSE_RED and not x1:     # (x1: the layer below takes its BN-backward sums from its own pass over dz, xyz1.hip)
                    pc = consts[l - 1]
                    fused_red = torch.empty(gemm_parts, 2, cin, device=dev, dtype=torch.float32)
                    nr = BwdRed()
                    nr.y = ys[l - 1].data_ptr()
                    nr.mean, nr.invstd, nr.scale, nr.shift = (pc[i].data_ptr() for i in range(4))
                    nr.red_partial = fused_red.data_ptr()
                    nr_ref = ctypes.byref(nr)
                if sparse_max and l == L - 1:
                    psel, wcat, hb, _ = max_prep
                    pc = consts[l - 1]
                    check(lib.papc_mlp_bwd_dx_max_f32(ptr(psel), ptr(argmax), spec.K, ys[l - 1].data_ptr(), cin, pc[2].data_ptr(),
                                                      pc[3].data_ptr(), ptr(wcat), ptr(hb), M, cin, cout, ptr(dz_prev), nr_ref, st),
                          "papc_mlp_bwd_dx_max_f32")
                else:
                    check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, cin, cout, ptr(dz_prev), None, nr_ref, st),
                          "papc_mlp_bwd_dx_f32")
                dz = dz_prev
            elif plain and ctx.x_needs_grad:
                wt = wts[l]
                grad_x = torch.empty(M, cin, device=dev, dtype=torch.float32)
                check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, cin, cout, ptr(grad_x), None, None, st), "papc_mlp_bwd_dx_f32")
            elif (not plain) and ctx.feats_needs_grad:
                wt = wts[l]
                grad_feats = _lib.zeros((spec.B, spec.N, spec.D), dev)
                sc = ScatterDst()
                sc.grad_feats, sc.idx = grad_feats.data_ptr(), ptr(idx)
                sc.N, sc.S, sc.K, sc.D = spec.N, spec.S, spec.K, spec.D
                sc.col0 = 3 if spec.xyz_first else 0
                check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, cin, cout, None, ctypes.byref(sc), None, st), "papc_mlp_bwd_dx_f32")
        for k0 in range(0, len(reduce_jobs), 8):       # (at most 8 jobs per launch)
            chunk = reduce_jobs[k0:k0 + 8]
            jobs = (ReduceJob * len(chunk))()
            for j, (part_t, nch, ld_, n1_, o1, n2_, o2, acc_) in zip(jobs, chunk):
                j.partial, j.n_chunks, j.accumulate, j.ld, j.n1, j.n2, j.out1, j.out2 = part_t.data_ptr(), nch, acc_, ld_, n1_, n2_, o1, o2
            check(lib.papc_reduce_partials_batch_f32(jobs, len(chunk), st), "papc_reduce_partials_batch_f32")
        return (None, None, None, None, grad_feats, None, grad_x) + tuple(grads)


def grad_targets_of(params):
    """Per parameter: its .grad tensor when the parameter opted in to in-place accumulation (a view of distributed.FlatParams.grad),
    else None -- or None altogether when no parameter did.  With a target the backward adds the gradient in place (one fused accumulate
    in the reduce kernels) and hands autograd ``None`` for it: no per-parameter AccumulateGrad add kernels.  Entries that are not
    opted-in leaf Parameters (e.g. the zero-padded view of a first conv weight, layers._pad_features) go back through autograd as usual."""
    tg = []
    for p in params:
        g = None
        # explicit opt-in (distributed.FlatParams marks its parameters): writing .grad behind autograd's back skips AccumulateGrad
        # hooks, so torch's DistributedDataParallel, post-accumulate hooks, torch.autograd.grad() and checkpoint recomputation would
        # miss or double-count these gradients -- a parameter that merely HAS a .grad (second step of any plain optimizer loop) does
        # not qualify
        if isinstance(p, torch.nn.Parameter) and getattr(p, "_papc_inplace_grad", False):
            g = p.grad
            if not (p.requires_grad and g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.shape == p.shape):
                g = None
        tg.append(g)
    return tg if any(t is not None for t in tg) else None


def shared_mlp_max(spec, bn_buffers, xyz, new_xyz, feats, idx, params, x_rows=None):
    # re-evaluated on every forward: a cached list would go stale when the optimizer replaces the .grad tensors
    # (e.g. zero_grad(set_to_none=True)) and gradients would silently land in orphaned buffers
    spec.grad_targets = grad_targets_of(params) if (torch.is_grad_enabled() and not spec.eval_bn) else None
    if _PY_ORCH or _SPARSE_MAX:       # the call sequences spelled out in Python (A/B against the library's own orchestration, csrc/sa_mlp.hip)
        from . import smallm
        if smallm.eligible(spec, xyz, feats, idx, x_rows, params):      # few rows, groups of 128 (group_all layers): csrc/smallm.hip
            return smallm.PlanesMLPMax.apply(spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, *params)
        return SharedMLPMax.apply(spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, *params)
    from .stack import SharedMLPStack
    return SharedMLPStack.apply(spec, bn_buffers, xyz, new_xyz, feats, idx, x_rows, *params)
