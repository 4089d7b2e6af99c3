"""Concatenations and transposed copies around the layers as ONE launch of the library's strided batch copy (csrc/bn_ops.hip,
papc_copy_strided_batch_f32) each way, instead of one library copy / cat kernel per piece.

Reference sites: ``paddle.concat`` of the MSG branches (PAPC/models/layers/pointnet2_basic_layers.py:280), of points1 and the interpolated
features (:326-327), of the one-hot label / coordinates / features ahead of fp1 (PAPC/models/segment/pointnet2/pointnet2.py:45), and the
``transpose`` of the feature tensor in front of a grouping layer (:205).  The inputs may be arbitrary strided views (transposes, expands);
the result is a fresh contiguous tensor; the backward hands every input a contiguous gradient of its logical shape.
"""
import torch

from . import _lib
from ._lib import CopyJob, check, stream_ptr


def _launch(jobs):
    lib = _lib.load()
    for k0 in range(0, len(jobs), 8):
        chunk = jobs[k0:k0 + 8]
        arr = (CopyJob * len(chunk))()
        for a, (src, soff, sst, dst, doff, dst_st, B, R, C) in zip(arr, chunk):
            a.src, a.dst = src.data_ptr() + 4 * soff, dst.data_ptr() + 4 * doff
            a.B, a.R, a.C = B, R, C
            a.sb, a.sr, a.sc = sst
            a.db, a.dr, a.dc = dst_st
        check(lib.papc_copy_strided_batch_f32(arr, len(chunk), stream_ptr()), "papc_copy_strided_batch_f32")


class _CatCopy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dim, *xs):
        B = xs[0].shape[0]
        other = 3 - dim                                   # the 3-D index that is not concatenated (dim is 1 or 2)
        n_other = xs[0].shape[other]
        sizes = [x.shape[dim] for x in xs]
        shape = [B, 0, 0]
        shape[dim], shape[other] = sum(sizes), n_other
        out = torch.empty(shape, device=xs[0].device, dtype=torch.float32)
        ost = out.stride()
        jobs, off = [], 0
        for x, n in zip(xs, sizes):
            jobs.append((x, 0, x.stride(), out, off * ost[dim], ost, B, x.shape[1], x.shape[2]))
            off += n
        _launch(jobs)
        ctx.dim, ctx.sizes = dim, sizes
        return out

    @staticmethod
    def backward(ctx, g):
        dim, sizes = ctx.dim, ctx.sizes
        gst = g.stride()
        B = g.shape[0]
        grads, jobs, off = [], [], 0
        for i, n in enumerate(sizes):
            if ctx.needs_input_grad[1 + i]:
                shape = list(g.shape)
                shape[dim] = n
                gi = torch.empty(shape, device=g.device, dtype=torch.float32)
                jobs.append((g, off * gst[dim], gst, gi, 0, gi.stride(), B, shape[1], shape[2]))
                grads.append(gi)
            else:
                grads.append(None)
            off += n
        if jobs:
            _launch(jobs)
        return (None,) + tuple(grads)


def cat_copy(xs, dim):
    """torch.cat(xs, dim) for float32 CUDA tensors [B, R, C] (dim 1 or 2; negative dims allowed), any strides -> contiguous [B, ., .]"""
    xs = list(xs)
    dim = dim % 3
    ok = dim in (1, 2) and all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.numel() > 0 for x in xs)
    if not ok:
        raise _lib.PapcError("cat_copy needs non-empty float32 CUDA tensors [B, R, C] and dim in {1, 2} (no CPU fallback)")
    return _CatCopy.apply(dim, *xs)


def contiguous_copy(x):
    """x.contiguous() for a float32 CUDA tensor [B, R, C] view (a transpose in front of a grouping layer) on the library's copy kernel;
    returns x itself when it already is contiguous"""
    if x.is_contiguous() and x.dtype == torch.float32:
        return x
    return cat_copy([x], 2)


class _PadCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, at, pad):
        B, R, C = x.shape
        out = torch.empty(B, R, C + pad, device=x.device, dtype=torch.float32)
        zero = _lib.const_zeros((1,), x.device)
        xs, os_ = x.stride(), out.stride()
        jobs = [(zero, 0, (0, 0, 0), out, at, os_, B, R, pad)]
        if at > 0:
            jobs.append((x, 0, xs, out, 0, os_, B, R, at))
        if at < C:
            jobs.append((x, at * xs[2], xs, out, at + pad, os_, B, R, C - at))
        _launch(jobs)
        ctx.at, ctx.pad = at, pad
        return out

    @staticmethod
    def backward(ctx, g):
        at, pad = ctx.at, ctx.pad
        B, R, Cp = g.shape
        C = Cp - pad
        gx = torch.empty(B, R, C, device=g.device, dtype=torch.float32)
        gs, xs = g.stride(), gx.stride()
        jobs = []
        if at > 0:
            jobs.append((g, 0, gs, gx, 0, xs, B, R, at))
        if at < C:
            jobs.append((g, (at + pad) * gs[2], gs, gx, at, xs, B, R, C - at))
        _launch(jobs)
        return gx, None, None


def pad_cols(x, at, pad):
    """x [B, R, C] (or [R, C]) with ``pad`` zero columns inserted in front of column ``at`` -> contiguous [.., C + pad]: the zero feature
    channels / weight columns that make a grouped layer's feature count a multiple of four (layers._pad_features); one launch each way"""
    if pad == 0:
        return x
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 3)):
        raise _lib.PapcError("pad_cols needs a float32 CUDA tensor [B, R, C] or [R, C] (no CPU fallback)")
    if x.dim() == 2:
        return _PadCols.apply(x.unsqueeze(0), at, pad).squeeze(0)
    return _PadCols.apply(x, at, pad)
