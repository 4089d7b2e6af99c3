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


class PointLists:
    """The inverse of a grouping (csrc/lingather.hip, papc_point_lists_f32): for every source point the physical rows that gathered it, ascending.
    prange [B*N, 2] int32 | prow [cap] int32 | pmeta [cap (+ 3 B N), 4] float32 (xyz_j - centre, multiplicity weight); ``compact`` = which row layout
    the lists index (the compacted plan's, or the padded [B,S,K] lists').  The 3 B N rows after the entries (``pmom``, a view; None when the tensor
    has no room for them) hold the lists' per-point moments (sum w | sum w d | sum w d d^T): with them the backward reads dz alone.  With the lists the gather-add first layer's backward (the only float-atomic
    kernel of a training step) becomes a segmented sum in fixed order: bit-reproducible gradients, and about half the time on MI355X."""
    __slots__ = ("prange", "prow", "pmeta", "compact", "pmom")

    def __init__(self, tensors, compact):
        self.prange, self.prow, self.pmeta = tensors
        self.compact = bool(compact)
        self.pmom = _moments_view(self.prange, self.prow, self.pmeta)

    def tensors(self):
        return (self.prange, self.prow, self.pmeta)


# "0": never (the float-atomic backward everywhere); "1": for compacted stacks only; "2" (default): for padded ones too.  Measured on MI355X (round 6,
# SA2 of the SSG classifier): compacted 76.7 + 5.5 (atomics + the fill of G) -> 62 us (lists) -> 41 us (lists, dz alone: papc_lingather_bwd_pp_f32).
# PADDED: with one entry per row the step was 0.41 ms SLOWER (every padding copy of a group's first neighbour sits in that one point's list: hundreds
# of rows on one wave, where the atomic kernel pre-sums them per group in a register); since the builder makes a group's copies ONE weighted entry
# (identical rows, identical gradients) the kernel takes 33 us against 85 + 5.5 and the padded step is 0-10 us faster -- the builder's time on the
# sampling stream eats most of it -- but, like the compacted one, free of float atomics: bit-reproducible.  Config 3: 5.56 -> 5.58 ms.
LISTS = int(os.environ.get("PAPC_POINT_LISTS", "2"))
MAX_LIST_POINTS = 8192                                      # source points per cloud the list builder holds in LDS


def _moments_view(prange, prow, pmeta):
    """the rows of pmeta after its ``cap`` entries: [B*N, 12] per-point moments, or None (a caller's tensor of exactly ``cap`` rows)"""
    cap, BN = prow.shape[0], prange.shape[0]
    return pmeta[cap:cap + 3 * BN].view(BN, 12) if pmeta.shape[0] >= cap + 3 * BN else None


def alloc_lists(B, N, G, K, device):
    cap = (G * K + 127) // 128 * 128
    return (torch.empty(B * N, 2, device=device, dtype=torch.int32), torch.empty(cap, device=device, dtype=torch.int32),
            torch.empty(cap + 3 * B * N, 4, device=device, dtype=torch.float32))


def point_lists(xyz, new_xyz, idx, cplan=None, out=None):
    """xyz [B,N,3] (any strides), new_xyz [B,S,3], idx [B,S,K] int32, ``cplan`` = the grouping's CompactPlan when its stack runs compacted
    -> PointLists; ``out`` = optional preallocated tensors (``alloc_lists``) the kernel writes into."""
    B, S, K = idx.shape
    N = xyz.shape[1]
    t = out if out is not None else alloc_lists(B, N, B * S, K, idx.device)
    g = _lib.GroupSrc()
    g.xyz, g.sb, g.sn, g.sc = xyz.data_ptr(), xyz.stride(0), xyz.stride(1), xyz.stride(2)
    g.new_xyz, g.idx = new_xyz.data_ptr(), idx.data_ptr()
    g.N, g.S, g.K = N, S, K
    start = None
    if cplan is not None:
        g.cidx, g.seg_grp, g.rows_dev, g.wstat = cplan.cidx.data_ptr(), cplan.seg_grp.data_ptr(), cplan.rows.data_ptr(), cplan.wrow.data_ptr()
        start = cplan.start.data_ptr()
    pl = PointLists(t, cplan is not None)
    check(_lib.load().papc_point_lists_f32(ctypes.byref(g), B, start, *(ptr(x) for x in t), ptr(pl.pmom) if pl.pmom is not None else None, stream_ptr()),
          "papc_point_lists_f32")
    return pl


def stack_ok(M, K, couts):
    """whether a gather-add-first stack of these output widths has its compacted kernel flavours (papc_mlp_compact_ok)"""
    if POLICY == "0":
        return False
    arr = (ctypes.c_int * len(couts))(*couts)
    return bool(_lib.load().papc_mlp_compact_ok(M, K, len(couts), arr))
