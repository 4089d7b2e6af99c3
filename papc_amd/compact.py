"""Compacted grouping: the distinct neighbours of every ball-query list, on the device (csrc/compact.hip).

``query_ball_point`` pads each neighbourhood to ``nsample`` slots with copies of its first hit
(/root/reference/PAPC/models/layers/pointnet2_basic_layers.py:118-124).  Copies are identical rows through the whole conv / BN / ReLU stack
(:214-217), do not change the max over the neighbourhood (:219), and enter the train-mode BatchNorm statistics and the backward sums only
through their multiplicity -- so the stack can run on the distinct rows plus one weight per group and give the same function with the same
gradients.  On the benchmark's clouds that halves SA2's rows (nsample 64 at radius 0.4 over 512 points: 33 distinct neighbours on average).

A plan is seven small device tensors made from the [B, S, K] index lists; the row count stays in device memory (``rows[0]``), so a captured
training step replays with whatever the next batch's clouds give.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

POLICY = os.environ.get("PAPC_COMPACT", "auto")      # "0": never, "1": wherever the kernels allow, "auto": where it pays (measured once per layer)
AUTO_MAX_FRACTION = 0.75                             # auto: compact when the physical rows are at most this fraction of B * S * nsample


class CompactPlan:
    """cnt8 [G] scratch | start [G+1] | rows [2] = (physical rows rounded up to 128, exact) | cidx [cap] | seg_grp [cap/8] | wrow [cap] | coef [G]"""
    __slots__ = ("cnt8", "start", "rows", "cidx", "seg_grp", "wrow", "coef", "G", "K")

    def __init__(self, tensors, G, K):
        self.cnt8, self.start, self.rows, self.cidx, self.seg_grp, self.wrow, self.coef = tensors
        self.G, self.K = G, K

    def tensors(self):
        return (self.cnt8, self.start, self.rows, self.cidx, self.seg_grp, self.wrow, self.coef)

    def fraction(self):
        """physical rows / padded rows (synchronises: diagnostics and the one-off auto decision only)"""
        return float(self.rows[0].item()) / float(self.G * self.K)


def alloc(G, K, device):
    cap = (G * K + 127) // 128 * 128              # (the last group takes the tail up to a multiple of 128 rows)
    i32 = dict(device=device, dtype=torch.int32)
    return (torch.empty(G, **i32), torch.empty(G + 1, **i32), torch.empty(2, **i32), torch.empty(cap, **i32), torch.empty(cap // 8, **i32),
            torch.empty(cap, device=device, dtype=torch.float32), torch.empty(G, device=device, dtype=torch.float32))


def plan(idx, out=None):
    """idx [B, S, K] int32 (ball-query lists) -> CompactPlan; ``out`` = optional preallocated tensors (``alloc``) the kernels write into."""
    B, S, K = idx.shape
    G = B * S
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.is_cuda
    t = out if out is not None else alloc(G, K, idx.device)
    check(_lib.load().papc_compact_plan_f32(ptr(idx), G, K, *(ptr(x) for x in t), stream_ptr()), "papc_compact_plan_f32")
    return CompactPlan(t, G, K)


def stack_ok(M, K, couts):
    """whether a gather-add-first stack of these output widths has its compacted kernel flavours (papc_mlp_compact_ok)"""
    if POLICY == "0":
        return False
    arr = (ctypes.c_int * len(couts))(*couts)
    return bool(_lib.load().papc_mlp_compact_ok(M, K, len(couts), arr))
