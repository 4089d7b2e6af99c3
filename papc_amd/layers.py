"""PointNet++ set-abstraction layers with the reference's constructor and forward signatures.

Mirrors /root/reference/PAPC/models/layers/pointnet2_basic_layers.py: ``PointNetSetAbstraction`` :179-221 and
``PointNetSetAbstractionMsg`` :224-281, ``PointNetFeaturePropagation`` :284-335.  forward(xyz [B,3,N], points [B,D,N] | None) -> (new_xyz [B,3,S],
new_points [B,D',S]).  The returned tensors are transposed *views* of point-major buffers ([B,S,3], [B,S,D']),
so chaining layers costs no transposes: the next layer's ``.transpose(1,2)`` (:203-205) is contiguous again.

Deliberate, flagged deviations from the source:
  * the conv/BN parameters are registered (``nn.ModuleList``) and trained; the source keeps them in plain Python
    lists (:185-191, :230-241) so its optimiser never sees them;
  * gradients flow through the gathers; the source cuts autograd at every ``index_points`` (:57-60).
  ``reference_quirks=True`` restores both behaviours (SA parameters frozen, gather gradients cut).
  * as in the source, the SA BatchNorms always normalise with batch statistics (``.eval()`` never reaches them).
"""
import os

import torch
import torch.nn as nn

from . import _lib
from . import functional as F_
from .copyops import cat_copy, contiguous_copy, pad_cols
from .mlp import StackSpec, shared_mlp_max


def _stack_params(convs, bns):
    ps = []
    for conv, bn in zip(convs, bns):
        ps += [conv.weight, conv.bias, bn.weight, bn.bias]
    return ps


def _bn_buffers(bns):
    return [(bn.running_mean, bn.running_var) for bn in bns]


def _pad_features(feats, params, xyz_first, feats_padded=None):
    """The gather kernels move features as float4: a feature count that is not a multiple of 4 (e.g. the 3 normal /
    xyz channels the segmentation nets feed to SA1) would fall to the element-wise path, ~10x slower.  Pad the features with
    zero channels and the first conv weight with matching zero columns instead (a view-level change: same result, the
    gradient flows back to the real columns through the concatenation)."""
    D = feats.shape[2] if feats_padded is None else feats_padded[1]
    pad = (-D) % 4
    if pad == 0:
        return feats, params, D
    if feats_padded is None:
        feats = pad_cols(feats, D, pad)
    else:
        feats = feats_padded[0]               # (padded once by the caller for several stacks: the MSG branches)
    w = params[0]
    w2 = w.reshape(w.shape[0], -1)
    # columns [xyz(3), feats(D)] -> [xyz, feats, 0]   or   [feats(D), xyz(3)] -> [feats, 0, xyz]
    w_pad = pad_cols(w2, w2.shape[1] if xyz_first else D, pad)
    return feats, [w_pad] + list(params[1:]), D + pad


class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, reference_quirks=False, init_dist=1.0):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))      # :189
            self.mlp_bns.append(nn.BatchNorm2d(out_channel, eps=1e-5))          # :190 (paddle default epsilon)
            last_channel = out_channel
        self.group_all = group_all
        self.reference_quirks = reference_quirks
        self.init_dist = init_dist   # the source initialises the FPS running distance with ones (:75)
        # compacted stack (compact.py): None = decide once from the data (the first sampling outside a graph capture measures how many of the
        # nsample slots are padding copies), True / False = forced
        self.compact = None
        self._compact_on = None
        if reference_quirks:
            for p in self.parameters():
                p.requires_grad_(False)

    def _compact_mode(self, B):
        """None: this stack has no compacted flavour (or PAPC_COMPACT=0); True / False: decided; "probe": to be measured"""
        from . import compact as C
        convs = self.mlp_convs
        D = convs[0].in_channels - 3
        if self.group_all or self.compact is False or D < 16 or D % 4 or len(convs) < 2:
            return None
        if not C.stack_ok(B * self.npoint * self.nsample, self.nsample, [c.out_channels for c in convs]):
            return None
        if self.compact is True or C.POLICY == "1":
            return True
        return "probe" if self._compact_on is None else self._compact_on

    def _compact_plan(self, idx, out=None):
        """the compacted layout of these ball-query lists, or None when the layer runs padded"""
        from . import compact as C
        mode = self._compact_mode(idx.shape[0])
        if mode is None or mode is False:
            return None
        if mode == "probe":
            if _lib._capturing():          # undecided inside a capture: stay on the padded path (a decision needs a host read)
                return None
            cp = C.plan(idx, out)
            self._compact_on = cp.fraction() <= C.AUTO_MAX_FRACTION
            return cp if self._compact_on else None
        return C.plan(idx, out)

    def sample_fps(self, xyz, start_idx=None, out=None):
        """Farthest-point sampling alone (:143-144): xyz [B,3,N] -> new_xyz [B,S,3] (written into ``out`` when given).  The serial half of
        :meth:`sample` -- npoint dependent argmax steps on B workgroups -- which a training loop can run TWO batches ahead, beside the rest of
        the previous batch's pyramid (bench.py: the split pyramid)."""
        if self.group_all:
            return None
        xyz = xyz.transpose(1, 2)
        if xyz.dtype != torch.float32:
            xyz = xyz.float()
        return F_._fps_raw(xyz, self.npoint, start_idx, self.init_dist, new_xyz_out=out)[1]

    def sample(self, xyz, start_idx=None, out=None, new_xyz=None):
        """The weight-independent half of the layer (FPS + ball query, :143-145) on its own: xyz [B,3,N] ->
        (new_xyz [B,S,3], idx [B,S,K] int32).  Lets a training loop run batch i+1's sampling on a side stream while
        batch i's MLP kernels own the other CUs (FPS is a serial chain that occupies only B of the 256 CUs).
        ``out`` = optional preallocated (new_xyz, idx) the kernels write straight into (no copies in a captured step);
        ``new_xyz`` = the centroids :meth:`sample_fps` already made for this xyz (no FPS here)."""
        if self.group_all:
            return None
        xyz = xyz.transpose(1, 2)
        if xyz.dtype != torch.float32:
            xyz = xyz.float()
        skip = _diag_skip(self.npoint) if out is not None else ()      # (timing diagnostics: stages left out, their buffers stale)
        if new_xyz is None:
            if "fps" in skip:
                new_xyz = out[0]
            else:
                _, new_xyz = F_._fps_raw(xyz, self.npoint, start_idx, self.init_dist, new_xyz_out=None if out is None else out[0])
        if "bq" in skip:
            idx = out[1]
        else:
            idx = F_._ball_query_raw([self.radius], [self.nsample], xyz, new_xyz, outs=None if out is None else [out[1]])[0]
        if self._xyz_first(xyz.shape[0]):
            # coordinates-only stack whose first layer runs through its input moments: the grouped centred coordinates and their
            # moments are weight-independent too (mlp.xyz_pregroup) -- part of the plan
            from .mlp import xyz_pregroup
            if "xyzpre" in skip and len(out) >= 4:
                return new_xyz, idx, out[2], out[3]
            xc, gpart = xyz_pregroup(xyz, new_xyz, idx, out=None if (out is None or len(out) < 4) else (out[2], out[3]))
            return new_xyz, idx, xc, gpart
        if "compact" in skip and len(out) in (9, 12):
            return tuple(out)
        cp = self._compact_plan(idx, out=None if (out is None or len(out) not in (9, 12)) else tuple(out[2:9]))
        # (new_xyz, idx[, cnt8, start, rows, cidx, seg_grp, wrow, coef][, prange, prow, pmeta]): the compact plan and the grouping's point
        # lists (the inverse index the gather-add backward sums over) are part of the plan like the lists themselves
        res = (new_xyz, idx) + (cp.tensors() if cp is not None else ())
        if "lists" in skip and out is not None and len(out) == len(res) + 3:
            return res + tuple(out[len(res):])
        if self._wants_lists(xyz, cp):
            from . import compact as _c
            have = out is not None and len(out) == len(res) + 3
            pl = _c.point_lists(xyz, new_xyz, idx, cp, out=tuple(out[len(res):]) if have else None)
            res = res + pl.tensors()
        return res

    def _wants_lists(self, xyz, cplan):
        """point lists are made for a training layer whose first layer can run as a gather-add (features present: decided by the stack plan) --
        by default only when the stack runs compacted (compact.LISTS)"""
        from . import compact as _c
        D = self.mlp_convs[0].in_channels - 3
        on = _c.LISTS >= 2 or (_c.LISTS == 1 and cplan is not None)
        return (on and self.training and not self.group_all and D >= 16 and D % 4 == 0 and len(self.mlp_convs) >= 2
                and xyz.shape[1] <= _c.MAX_LIST_POINTS and not self.reference_quirks)

    @staticmethod
    def _parse_plan(sampled, G, K):
        """(new_xyz, idx, xyz_pre, CompactPlan, PointLists) from a :meth:`sample` result"""
        from .compact import CompactPlan, PointLists
        new_xyz, idx = sampled[0], sampled[1]
        rest = tuple(sampled[2:])
        xyz_pre = rest if len(rest) == 2 else None
        cplan = plists = None
        if len(rest) in (7, 10):
            cplan = CompactPlan(rest[:7], G, K)
            rest = rest[7:]
        if len(rest) == 3:
            plists = PointLists(rest, cplan is not None)
        return new_xyz, idx, xyz_pre, cplan, plists

    def _xyz_first(self, B):
        from .mlp import xyz_first_layer_ok
        convs = self.mlp_convs
        return (not self.group_all and convs[0].in_channels == 3 and len(convs) >= 3
                and xyz_first_layer_ok(B * self.npoint * self.nsample, convs[0].out_channels, convs[1].out_channels, len(convs)))

    def forward(self, xyz, points, start_idx=None, sampled=None, wt_table=None):
        """xyz [B,3,N], points [B,D,N] or None -> new_xyz [B,3,S], new_points [B,D',S].
        ``sampled`` = optional (new_xyz, idx) from :meth:`sample` (skips FPS / ball query here);
        ``wt_table`` = optional result of ``mlp.precompute_wt`` covering this layer's weights, made for THIS forward pass."""
        xyz = xyz.transpose(1, 2)                                               # :203  [B,N,3] view
        if xyz.dtype != torch.float32:
            xyz = xyz.float()
        feats = None
        if points is not None:
            feats = contiguous_copy(points.float().transpose(1, 2))                 # :205  [B,N,D]
        B, N, _ = xyz.shape
        D = 0 if feats is None else feats.shape[2]
        if self.group_all:                                                      # sample_and_group_all :160-176
            S, K = 1, N
            new_xyz = _lib.const_zeros((B, 1, 3), xyz.device)                   # :170 (read-only: cached per device and batch size)
            idx = None
        else:                                                                   # sample_and_group :129-157
            S, K = self.npoint, self.nsample
            xyz_pre = None
            cplan = plists = None
            if sampled is not None:
                new_xyz, idx, xyz_pre, cplan, plists = self._parse_plan(sampled, B * S, K)
            else:
                _, new_xyz = F_._fps_raw(xyz, S, start_idx, self.init_dist)
                idx = F_._ball_query_raw([self.radius], [K], xyz, new_xyz)[0]
                if feats is not None:
                    cplan = self._compact_plan(idx)
                    if self._wants_lists(xyz, cplan) and torch.is_grad_enabled():
                        from . import compact as _c
                        plists = _c.point_lists(xyz, new_xyz, idx, cplan)
        params = _stack_params(self.mlp_convs, self.mlp_bns)
        if feats is not None:
            feats, params, D = _pad_features(feats, params, True)
        spec = StackSpec(B, N, S, K, D, xyz_first=True, eps=self.mlp_bns[0].eps, momentum=0.9,
                         cut_gather_grad=self.reference_quirks)
        spec.wt_table = wt_table
        if not self.group_all and feats is None:
            spec.xyz_pre = xyz_pre
        if not self.group_all and feats is not None:
            spec.compact = cplan
            spec.plists = plists
        out = shared_mlp_max(spec, _bn_buffers(self.mlp_bns), xyz, new_xyz, feats, idx, params)   # :214-219
        new_points = out.view(B, S, -1).transpose(1, 2)                         # [B,D',S]
        # (group_all: new_xyz is the cached READ-ONLY zero centre of sample_and_group_all, :170 -- a clone here would put a copy kernel
        # into every training step; callers must not edit the returned coordinates in place)
        return new_xyz.transpose(1, 2), new_points                              # :220-221


def _diag_skip(npoint):
    """PAPC_DIAG_SKIP=fps512,bq512,xyzpre512,fps128,bq128,compact128,lists128: stages of sample() to leave out when it writes into preallocated
    buffers (bench.py's sampling graph) -- what each stage costs the step; the plan is then STALE, results garbage.  Timing diagnostics only."""
    v = os.environ.get("PAPC_DIAG_SKIP")
    if not v:
        return ()
    suf = str(npoint)
    return tuple(t[:-len(suf)] for t in v.split(",") if t.endswith(suf))


MSG_BRANCH_STREAMS = os.environ.get("PAPC_MSG_STREAMS", "1") != "0"     # the MSG layers' radius branches on parallel streams
_BRANCH_STREAMS = {}


def set_branch_streams(model, on):
    """per-model switch for the MSG layers' parallel branch streams (None restores the process default); returns the previous settings"""
    old = []
    for m in model.modules():
        if isinstance(m, PointNetSetAbstractionMsg):
            old.append((m, m.branch_streams))
            m.branch_streams = on
    return old


def _branch_streams(dev, n):
    lst = _BRANCH_STREAMS.setdefault(str(dev), [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


class PointNetSetAbstractionMsg(nn.Module):
    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, reference_quirks=False, init_dist=1.0):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for i in range(len(mlp_list)):
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last_channel = in_channel + 3                                       # :235
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel, eps=1e-5))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
        self.reference_quirks = reference_quirks
        self.init_dist = init_dist
        self.compact = None            # compacted branches (compact.py): None = decide once per branch from the data, True / False = forced
        self._compact_on = {}
        self.branch_streams = None     # radius branches on parallel streams: None = the process default (PAPC_MSG_STREAMS), True / False = this layer
        if reference_quirks:
            for p in self.parameters():
                p.requires_grad_(False)

    def _branch_compact_mode(self, i, B):
        """None: branch i has no compacted flavour; True / False: decided; "probe": to be measured (see PointNetSetAbstraction)"""
        from . import compact as C
        convs = self.conv_blocks[i]
        D = convs[0].in_channels - 3
        Dp = D + ((-D) % 4)
        if self.compact is False or Dp < 16 or len(convs) < 2:
            return None
        if not C.stack_ok(B * self.npoint * self.nsample_list[i], self.nsample_list[i], [c.out_channels for c in convs]):
            return None
        if self.compact is True or C.POLICY == "1":
            return True
        on = self._compact_on.get(i)
        return "probe" if on is None else on

    def _branch_plan(self, i, idx, out=None):
        from . import compact as C
        mode = self._branch_compact_mode(i, idx.shape[0])
        if mode is None or mode is False:
            return None
        if mode == "probe":
            if _lib._capturing():
                return None
            cp = C.plan(idx, out)
            self._compact_on[i] = cp.fraction() <= C.AUTO_MAX_FRACTION
            return cp if self._compact_on[i] else None
        self._compact_on[i] = True      # forced (compact=True / PAPC_COMPACT=1): sample() and forward() read the branch's layout from this one flag
        return C.plan(idx, out)

    def _branch_wants_lists(self, i, N):
        """point lists (compact.point_lists) for a compacted branch of a training layer: its gather-add backward as a segmented sum"""
        from . import compact as C
        D = self.conv_blocks[i][0].in_channels - 3
        return (C.LISTS >= 1 and self.training and not self.reference_quirks and D + ((-D) % 4) >= 16 and len(self.conv_blocks[i]) >= 2
                and N <= C.MAX_LIST_POINTS)

    def sample(self, xyz, start_idx=None, out=None):
        """The weight-independent half (one FPS, one ball-query scan for all radii, :258-262) on its own: xyz [B,3,N] ->
        (new_xyz [B,S,3], idx_0 .. idx_{R-1} [B,S,K_r] int32, then per branch that runs compacted its 7 compact-plan tensors and -- for a
        training layer -- the 3 point-list tensors of that layout).  ``out`` = a previous result of this method for the same shapes: the kernels
        write into it."""
        xyz = xyz.transpose(1, 2)
        if xyz.dtype != torch.float32:
            xyz = xyz.float()
        R = len(self.radius_list)
        _, new_xyz = F_._fps_raw(xyz, self.npoint, start_idx, self.init_dist, new_xyz_out=None if out is None else out[0])
        idxs = F_._ball_query_raw(self.radius_list, self.nsample_list, xyz, new_xyz, outs=None if out is None else list(out[1:1 + R]))
        res = [new_xyz] + list(idxs)
        pos = 1 + R
        for i in range(R):
            forced = self._branch_compact_mode(i, xyz.shape[0]) is True
            have = out is not None and (forced or self._compact_on.get(i) is True) and len(out) >= pos + 7
            cp = self._branch_plan(i, idxs[i], out=tuple(out[pos:pos + 7]) if have else None)
            if cp is not None:
                res += list(cp.tensors())
                pos += 7
                if self._branch_wants_lists(i, xyz.shape[1]):
                    from . import compact as C
                    have_l = out is not None and len(out) >= pos + 3
                    res += list(C.point_lists(xyz, new_xyz, idxs[i], cp, out=tuple(out[pos:pos + 3]) if have_l else None).tensors())
                    pos += 3
        return tuple(res)

    def forward(self, xyz, points, start_idx=None, sampled=None):
        xyz = xyz.transpose(1, 2)                                               # :252
        if xyz.dtype != torch.float32:
            xyz = xyz.float()
        feats = None
        if points is not None:
            feats = contiguous_copy(points.float().transpose(1, 2))
        B, N, _ = xyz.shape
        D = 0 if feats is None else feats.shape[2]
        S = self.npoint
        R = len(self.radius_list)
        cplans = [None] * R
        plists = [None] * R
        if sampled is not None:
            from .compact import CompactPlan, PointLists
            new_xyz, idxs = sampled[0], list(sampled[1:1 + R])
            pos = 1 + R
            for i in range(R):
                if self._compact_on.get(i) is True and self._branch_compact_mode(i, B) is True and len(sampled) >= pos + 7:
                    cplans[i] = CompactPlan(tuple(sampled[pos:pos + 7]), B * S, self.nsample_list[i])
                    pos += 7
                    # (the lists are recognised by their shapes -- prange is [B*N, 2], pmeta [cap, 4] -- not by this layer's current mode: a plan
                    # made in another mode parses all the same)
                    if len(sampled) >= pos + 3 and sampled[pos].dim() == 2 and sampled[pos].shape[1] == 2 and sampled[pos + 2].dim() == 2:
                        plists[i] = PointLists(tuple(sampled[pos:pos + 3]), True)
                        pos += 3
        else:
            _, new_xyz = F_._fps_raw(xyz, S, start_idx, self.init_dist)             # :258 (one FPS for all radii)
            idxs = F_._ball_query_raw(self.radius_list, self.nsample_list, xyz, new_xyz)   # :260-262, one scan
            if feats is not None:
                cplans = [self._branch_plan(i, idxs[i]) for i in range(R)]
                if torch.is_grad_enabled():
                    from . import compact as C
                    plists = [C.point_lists(xyz, new_xyz, idxs[i], cplans[i]) if (cplans[i] is not None and self._branch_wants_lists(i, N)) else None
                              for i in range(R)]
        feats_in = feats
        padded = None
        if feats_in is not None and D % 4 != 0:          # (one padded copy of the features for all branches)
            padded = (pad_cols(feats_in, D, (-D) % 4), D)
        # The radius branches are independent until the concatenation (:264-280): branch i > 0 runs on its own stream (forked from / joined
        # into the caller's -- inside a hipGraph capture these become parallel graph branches), so one branch's launch-sized kernels
        # (BatchNorm finalizes, partial folds, the dependency gaps around them) run beside another branch's GEMMs.  Autograd replays each
        # branch's backward on the stream its forward ran on.
        par = (MSG_BRANCH_STREAMS if self.branch_streams is None else self.branch_streams) and xyz.is_cuda and R > 1
        side = _branch_streams(xyz.device, R - 1) if par else None
        main = torch.cuda.current_stream(xyz.device) if par else None

        def branch(i, K):
            params = _stack_params(self.conv_blocks[i], self.bn_blocks[i])
            Dp, feats_i = D, feats
            if feats_in is not None:
                feats_i, params, Dp = _pad_features(feats_in, params, False, padded)
            spec = StackSpec(B, N, S, K, Dp, xyz_first=False, eps=self.bn_blocks[i][0].eps, momentum=0.9,
                             cut_gather_grad=self.reference_quirks)             # feats first, then xyz (:267)
            if feats_in is not None:
                spec.compact = cplans[i]
                spec.plists = plists[i]
            o = shared_mlp_max(spec, _bn_buffers(self.bn_blocks[i]), xyz, new_xyz, feats_i, idxs[i], params)   # :271-276
            return o.view(B, S, -1)

        outs = [None] * R
        for i, K in enumerate(self.nsample_list):
            if par and i > 0:
                side[i - 1].wait_stream(main)
                # every tensor the branch reads was allocated on the caller's stream: tell the allocator the side stream uses it too, in the
                # forward here and in the backward autograd replays on the same stream (a freed block is otherwise handed to the next
                # main-stream allocation while a side-stream kernel may still be reading it)
                shared = [xyz, new_xyz, idxs[i], feats_in, None if padded is None else padded[0]]
                if cplans[i] is not None:
                    shared += list(cplans[i].tensors())
                if plists[i] is not None:
                    shared += list(plists[i].tensors())
                for t in shared:
                    if t is not None:
                        t.record_stream(side[i - 1])
                with torch.cuda.stream(side[i - 1]):
                    outs[i] = branch(i, K)
                outs[i].record_stream(main)     # (consumed by the concatenation on the caller's stream)
            else:
                outs[i] = branch(i, K)
        if par:
            for s_ in side:
                main.wait_stream(s_)
        new_points_concat = cat_copy(outs, 2).transpose(1, 2)                   # :280  [B,D',S]
        return new_xyz.transpose(1, 2), new_points_concat


class PointNetFeaturePropagation(nn.Module):
    """:284-335.  forward(xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D1,N] | None, points2 [B,D2,S]) -> [B,D',N].

    ``neighbours``:
      * ``"reference"`` (default) -- what the source computes: it sorts ``dists`` and THEN argsorts the sorted
        matrix (:316-317), so its ``idx`` is 0,1,2 for every query: the three smallest distances weight the
        support points 0, 1 and 2.  Kept bit-compatible so a PAPC checkpoint / loss curve carries over.
      * ``"nearest"`` -- the PointNet++ paper's intent: the weights go to the three nearest support points.
    Same flagged deviations as the SA layers: parameters are registered and trained, and the gradient flows into
    ``points2`` through the interpolation (the source's ``index_points`` cuts it, :57-60);
    ``reference_quirks=True`` restores both.  The returned tensor is a transposed view of a point-major buffer.
    """

    def __init__(self, in_channel, mlp, neighbours="reference", reference_quirks=False):
        super().__init__()
        assert neighbours in ("reference", "nearest")
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))      # :291
            self.mlp_bns.append(nn.BatchNorm1d(out_channel, eps=1e-5))          # :292
            last_channel = out_channel
        self.neighbours = neighbours
        self.reference_quirks = reference_quirks
        if reference_quirks:
            for p in self.parameters():
                p.requires_grad_(False)

    def plan(self, xyz1, xyz2, out=None):
        """The weight-independent part of the layer -- the 3-NN search (:315-322) -- on its own: xyz1 [B,3,N], xyz2 [B,3,S] ->
        (dist3, idx3, w3) or None when S == 1.  A training loop can compute it with the sampling pyramid (models.*.plan_sampling)."""
        if xyz2.shape[2] == 1:
            return None
        return F_.three_nn(xyz1.transpose(1, 2), xyz2.transpose(1, 2), out=out)

    def interpolate(self, xyz1, xyz2, points2, planned=None):
        """:311-323 on point-major tensors: xyz1 [B,N,3], xyz2 [B,S,3], points2 [B,S,D] -> [B,N,D]"""
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if self.reference_quirks:
            points2 = points2.detach()
        if S == 1:
            return points2.expand(B, N, points2.shape[2])                        # paddle.tile (:312)
        _, idx3, w3 = planned if planned is not None else F_.three_nn(xyz1, xyz2)
        first3 = self.neighbours == "reference"
        if first3:
            idx3 = _lib.const_idx3(B, N, idx3.device)                                 # argsort of a sorted row: 0, 1, 2 (cached constant)
        return F_.three_interpolate(points2, idx3, w3, first3=first3)

    def forward(self, xyz1, xyz2, points1, points2, planned=None):
        xyz1 = xyz1.transpose(1, 2)                                              # :305-306
        xyz2 = xyz2.transpose(1, 2)
        points2 = points2.transpose(1, 2)                                        # :308
        B, N, _ = xyz1.shape
        interpolated = self.interpolate(xyz1, xyz2, points2, planned)
        if points1 is not None:
            new_points = cat_copy([points1.transpose(1, 2).float(), interpolated], 2)         # :326-327
        else:
            new_points = interpolated
        rows = new_points.reshape(B * N, new_points.shape[2])
        if not rows.is_contiguous():
            rows = rows.contiguous()
        spec = StackSpec(B, N, N, 1, rows.shape[1], True, eps=self.mlp_bns[0].eps, momentum=0.9, pool=False)
        out = shared_mlp_max(spec, _bn_buffers(self.mlp_bns), None, None, None, None,
                             _stack_params(self.mlp_convs, self.mlp_bns), x_rows=rows)   # :331-333
        return out.view(B, N, -1).transpose(1, 2)
