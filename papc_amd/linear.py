"""A plain per-row Linear (1x1 conv without norm or activation) on this library's kernels -- the last layer of the part-segmentation
heads, ``conv2`` of /root/reference/PAPC/models/segment/pointnet2/pointnet2.py:49 (logits [B*N, num_parts] from 128 channels).

Forward = papc_mlp_gemm_f32 on plain rows.  The backward reuses the BN-aware row kernels with constants that make their dY the
upstream gradient itself (scale 1, shift huge -> the ReLU mask is always on; mean 0, invstd 1, c1 = c2 = 0): dW and dX are the stack
kernels, the bias gradient is the column sum those kernels' BN-backward reduction produces.  Parameters that opted in to in-place
accumulation (distributed.FlatParams) get their gradients added in place (no AccumulateGrad kernels).
"""
import ctypes

import torch

from . import _lib
from ._lib import BwdDy, check, ptr, stream_ptr


class _LinearRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, targets, rows, w, b):
        lib = _lib.load()
        st = stream_ptr()
        M, cin = rows.shape
        cout = w.shape[0]
        y = torch.empty(M, cout, device=rows.device, dtype=torch.float32)
        check(lib.papc_mlp_gemm_f32(0, ptr(rows), cin, None, None, None, ptr(w), ptr(b), M, cin, cout, ptr(y), None, None, st), "papc_mlp_gemm_f32")
        ctx.save_for_backward(rows, w, y)
        ctx.targets = targets
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gout):
        from .mlp import _dw_rows_per_chunk
        lib = _lib.load()
        st = stream_ptr()
        rows, w, y = ctx.saved_tensors
        M, cin = rows.shape
        cout = w.shape[0]
        dev = rows.device
        gout = gout.contiguous().float()
        one, zero, big = _lib.const_vec(1.0, cout, dev), _lib.const_vec(0.0, cout, dev), _lib.const_vec(1e30, cout, dev)
        dy = BwdDy()
        dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = 0, gout.data_ptr(), None, None, 1
        dy.y = y.data_ptr()
        dy.mean, dy.invstd, dy.scale, dy.shift = zero.data_ptr(), one.data_ptr(), one.data_ptr(), big.data_ptr()
        dy.c1, dy.c2 = zero.data_ptr(), zero.data_ptr()
        tg = ctx.targets
        tw = tg[0] if tg is not None else None
        tb = tg[1] if (tg is not None and ctx.has_bias) else None
        # dW = gout^T . rows
        rpc = _dw_rows_per_chunk(M, cout, cin)
        n_chunks = (M + rpc - 1) // rpc
        pld = cout * cin + cout
        part = torch.empty(n_chunks, pld, device=dev, dtype=torch.float32)
        check(lib.papc_mlp_bwd_dw_f32(ctypes.byref(dy), 0, ptr(rows), cin, None, None, None, M, cin, cout, rpc, part.data_ptr(),
                                      part.data_ptr() + 4 * cout * cin, pld, st), "papc_mlp_bwd_dw_f32")
        dw = tw.view(cout, cin) if tw is not None else torch.empty(cout, cin, device=dev, dtype=torch.float32)
        scratch = torch.empty(cout, device=dev, dtype=torch.float32)     # (the dW kernel's bias slot: the exact 0 of a BN-fed bias; unused here)
        check(lib.papc_reduce_partials2_f32(ptr(part), n_chunks, pld, cout * cin, ptr(dw), cout, ptr(scratch), 0, st) if tw is None else
              lib.papc_reduce_partials_strided_f32(ptr(part), n_chunks, pld, cout, cin, ptr(dw), cin, 1, st), "papc_reduce_partials")
        # db = column sums of gout: the BN-backward reduction with the always-on mask
        db = None
        if ctx.has_bias:
            n_parts = min(512, (M + 127) // 128)
            red = torch.empty(n_parts, 2, cout, device=dev, dtype=torch.float32)
            check(lib.papc_bn_bwd_reduce_f32(0, ptr(gout), None, None, 1, ptr(y), zero.data_ptr(), one.data_ptr(), one.data_ptr(), big.data_ptr(),
                                             M, cout, n_parts, ptr(red), st), "papc_bn_bwd_reduce_f32")
            dgam = torch.empty(cout, device=dev, dtype=torch.float32)
            c12 = torch.empty(2, cout, device=dev, dtype=torch.float32)
            db = tb if tb is not None else torch.empty(cout, device=dev, dtype=torch.float32)
            check(lib.papc_bn_bwd_finalize_f32(ptr(red), n_parts, M, cout, ptr(dgam), ptr(db), c12[0].data_ptr(), c12[1].data_ptr(),
                                               2 | (1 if tb is not None else 0), st), "papc_bn_bwd_finalize_f32")
        dx = None
        if ctx.needs_input_grad[1]:
            wt = torch.empty(cin, cout, device=dev, dtype=torch.float32)
            check(lib.papc_copy2d_f32(ptr(w), cin, ptr(wt), cout, cout, cin, 1, st), "papc_copy2d_f32")
            dx = torch.empty(M, cin, device=dev, dtype=torch.float32)
            check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, cin, cout, ptr(dx), None, None, st), "papc_mlp_bwd_dx_f32")
        return None, dx, (None if tw is not None else dw), (None if (tb is not None or db is None) else db)


def linear_rows(rows, weight, bias):
    """rows [M, Cin] @ weight[Cout, Cin]^T + bias -> [M, Cout], differentiable, on libpapc_hip.so only."""
    from .mlp import grad_targets_of
    if not rows.is_cuda:
        raise _lib.PapcError("linear_rows needs CUDA (ROCm) tensors: there is no CPU fallback")
    w2 = weight.reshape(weight.shape[0], -1)
    tg = None
    if torch.is_grad_enabled() and w2.data_ptr() == weight.data_ptr():
        tg = grad_targets_of([weight] + ([bias] if bias is not None else []))
        if tg is not None and (tg[0] is None or (bias is not None and tg[1] is None)):
            tg = None
    rows = rows if (rows.is_contiguous() and rows.dtype == torch.float32) else rows.contiguous().float()
    return _LinearRows.apply(tg, rows, w2, bias)
