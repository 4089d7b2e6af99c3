"""The callers of the hot path, same constructors and forward contracts as the reference models.

  PointNet2_SSG_Clas / PointNet2_MSG_Clas  <- /root/reference/PAPC/models/classify/pointnet2/pointnet2.py:6-41, :43-75
  PointNet_Basic_Clas                      <- /root/reference/PAPC/models/classify/pointnet_base/pointnet_base.py:4-47
  PointNet2_SSG_Seg / PointNet2_MSG_Seg    <- /root/reference/PAPC/models/segment/pointnet2/pointnet2.py:6-52, :54-100

Inputs are ``[B,3,N]`` float32 (``[B,6,N]`` with normals).  Everything runs in libpapc_hip.so, in train AND eval mode, the FC heads included
(head.py); the nn.Linear / BatchNorm1d / Dropout modules are the parameter holders.  CPU tensors raise PapcError (there is no CPU path); the
module chain remains only for TRAIN-mode head shapes outside the kernels' limits (more than 256 rows, widths that are no multiple of 4).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import (PointNetFeaturePropagation, PointNetSetAbstraction, PointNetSetAbstractionMsg, _bn_buffers,
                     _stack_params)
from . import _lib
from . import head as _head
from .copyops import cat_copy
from .mlp import StackSpec, shared_mlp_max

import os
_FUSED_HEAD = os.environ.get("PAPC_NO_FUSED_HEAD") != "1"     # A/B switch: fused classifier head (head.py) vs the library-op chain


def _starts(start_idx, n):
    if start_idx is None:
        return [None] * n
    assert len(start_idx) == n, "start_idx must hold one [B] tensor per FPS level"
    return list(start_idx)


def _bn1d(bn, y):
    """BatchNorm1d of the fallback head path with the convention of the fused kernels (and of paddle.nn.BatchNorm1D, which the
    reference uses: classify/pointnet2/pointnet2.py:18,21): in training the BIASED batch variance goes into the running estimate --
    torch's module would store the unbiased one, so the running statistics (and an exported .pdparams) would depend on which path a
    batch size selects."""
    if not (bn.training and bn.track_running_stats and y.shape[0] > 1):
        return bn(y)
    mean = y.mean(0)
    var = y.var(0, unbiased=False)
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    with torch.no_grad():
        bn.running_mean.mul_(1.0 - mom).add_(mean.detach(), alpha=mom)
        bn.running_var.mul_(1.0 - mom).add_(var.detach(), alpha=mom)
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
    return (y - mean) / torch.sqrt(var + bn.eps) * bn.weight + bn.bias


class _ClasHead:
    """fc1/bn1/drop1/fc2/bn2/drop2/fc3 (pointnet2.py:37-39): fused launches (head.py) in train and eval mode."""

    def _head(self, x, labels=None):
        """logits, or with ``labels`` (int64 [B]) the pair (mean softmax cross-entropy, logits): train.py:106-109's loss computed by the head's own
        launch (head.classifier_head_loss); the logits then carry no gradient."""
        spec = self.__dict__.get("_head_spec")
        if spec is None:
            spec = self.__dict__["_head_spec"] = _head.HeadSpec()
        if not x.is_cuda:
            raise _lib.PapcError("the classifier head needs CUDA (ROCm) tensors: there is no CPU fallback")
        if _FUSED_HEAD and _head.usable(x, self.fc1, self.fc2, self.fc3, self.training):
            if labels is not None:
                return _head.classifier_head_loss(spec, x, labels, self.fc1, self.bn1, self.drop1, self.fc2, self.bn2, self.drop2, self.fc3,
                                                  training=self.training)
            return _head.classifier_head(spec, x, self.fc1, self.bn1, self.drop1, self.fc2, self.bn2, self.drop2, self.fc3, training=self.training)
        # (train-mode shapes outside the kernels' limits, or PAPC_NO_FUSED_HEAD=1: the modules)
        x = self.drop1(F.relu(_bn1d(self.bn1, self.fc1(x))))
        x = self.drop2(F.relu(_bn1d(self.bn2, self.fc2(x))))
        logits = self.fc3(x)
        if labels is not None:
            return (_head.softmax_cross_entropy(logits, labels) if logits.is_cuda else F.cross_entropy(logits, labels.reshape(-1))), logits.detach()
        return logits


class PointNet2_SSG_Clas(nn.Module, _ClasHead):
    def __init__(self, name_scope='PointNet2_SSG_Clas_', num_classes=16, normal_channel=False, reference_quirks=False):
        super().__init__()
        in_channel = 6 if normal_channel else 3
        self.normal_channel = normal_channel
        q = dict(reference_quirks=reference_quirks)
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=in_channel, mlp=[64, 64, 128],
                                          group_all=False, **q)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256],
                                          group_all=False, **q)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3,
                                          mlp=[256, 512, 1024], group_all=True, **q)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, num_classes)

    def plan_sampling(self, inputs, start_idx=None, out=None, stage=None):
        """The whole weight-independent sampling pyramid of one batch (FPS1, ball query 1, FPS2, ball query 2): returns
        ((new_xyz1, idx1), (new_xyz2, idx2)).  Pass it to forward(plan=...); a training loop can compute it for batch
        i+1 on a side stream while batch i trains (see bench.py).  ``out`` = optional preallocated plan (same structure) that the
        kernels fill in place.  ``stage`` (needs ``out``): "fps1" = only the first level's farthest-point sampling, into out[0][0] (returns
        that tensor); "rest" = everything else, from the centroids out[0][0] already holds -- the two halves of a pyramid that a loop runs
        on two streams, the serial FPS chain one batch further ahead than the rest."""
        xyz = torch.as_tensor(inputs)
        if self.normal_channel:
            xyz = xyz[:, :3, :]
        s = _starts(start_idx, 2)
        with torch.no_grad():
            o = out if out is not None else (None, None)
            if stage == "fps1":
                return self.sa1.sample_fps(xyz, s[0], out=o[0][0])
            p1 = self.sa1.sample(xyz, s[0], out=o[0], new_xyz=o[0][0] if stage == "rest" else None)
            p2 = self.sa2.sample(p1[0].transpose(1, 2), s[1], out=o[1])
        return p1, p2

    def forward(self, inputs, start_idx=None, plan=None, after_sa2=None, tap=None, after_sa3=None, labels=None, after_sa1=None):
        """inputs [B,3,N]; ``labels`` = optional int64 [B]: return (loss, logits) instead of logits, the cross-entropy of train.py:106-109
        computed by the head's own launch; ``start_idx`` = optional (s1 [B], s2 [B]) FPS start indices (the source draws them at random);
        ``plan`` = optional result of :meth:`plan_sampling` for these inputs; ``after_sa2`` = optional callable invoked
        once SA2's kernels are enqueued -- from there to the end of SA3's backward only small-grid kernels run (group_all
        layer, FC head), the window in which a side stream can sample the next batch on otherwise idle CUs;
        ``tap`` = optional dict that receives ``l2_points``, the tensor SA3 consumes: a data-parallel loop runs the backward in
        two stages around it (head + SA3 first, their gradient bucket all-reducing while SA2 / SA1 follow; see bench.py)."""
        xyz = torch.as_tensor(inputs)
        B = xyz.shape[0]
        if self.normal_channel:
            norm, xyz = xyz[:, 3:, :], xyz[:, :3, :]
        else:
            norm = None
        s = _starts(start_idx, 2)
        pl = plan if plan is not None else (None, None)
        wt = None
        if torch.is_grad_enabled() and xyz.is_cuda:
            # the W^T operands of the three stacks' dX GEMMs in one launch (SA1 / SA2: layers 2, 3; SA3 also its first layer, whose
            # input features carry a gradient)
            from . import smallm
            from .mlp import precompute_wt
            ws = [c.weight for c in list(self.sa1.mlp_convs)[1:]] + [c.weight for c in self.sa2.mlp_convs]   # (SA2's first layer: the feature block of its W^T, gather-add backward)
            if not (smallm.takes_group_all(B, self.sa2.npoint, [c.weight.shape[0] for c in self.sa3.mlp_convs])):
                ws += [c.weight for c in self.sa3.mlp_convs]    # (the planes path builds its own W^T planes: nothing to transpose for it)
            wt = precompute_wt(ws, feat_blocks=[(self.sa2.mlp_convs[0].weight, True)])      # (SA2's gather-add first layer: its feature block, contiguous)
        l1_xyz, l1_points = self.sa1(xyz, norm, s[0], sampled=pl[0], wt_table=wt)
        if after_sa1 is not None:
            after_sa1()
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, s[1], sampled=pl[1], wt_table=wt)
        if after_sa2 is not None:
            after_sa2()
        if tap is not None:
            tap["l2_points"] = l2_points
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points, wt_table=wt)
        if after_sa3 is not None:
            after_sa3()
        x = l3_points.reshape(B, 1024)
        return self._head(x, labels)


class PointNet2_MSG_Clas(nn.Module, _ClasHead):
    def __init__(self, name_scope='PointNet2_MSG_Clas_', num_classes=16, normal_channel=False, reference_quirks=False):
        super().__init__()
        in_channel = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        q = dict(reference_quirks=reference_quirks)
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], in_channel,
                                             [[32, 32, 64], [64, 64, 128], [64, 96, 128]], **q)
        self.sa2 = PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320,
                                             [[64, 64, 128], [128, 128, 256], [128, 128, 256]], **q)
        self.sa3 = PointNetSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True, **q)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.5)
        self.fc3 = nn.Linear(256, num_classes)

    def forward(self, inputs, start_idx=None):
        xyz = torch.as_tensor(inputs)
        B = xyz.shape[0]
        if self.normal_channel:
            norm, xyz = xyz[:, 3:, :], xyz[:, :3, :]
        else:
            norm = None
        s = _starts(start_idx, 2)
        l1_xyz, l1_points = self.sa1(xyz, norm, s[0])
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, s[1])
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.reshape(B, 1024)
        return self._head(x)


class PointNet_Basic_Clas(nn.Module):
    """Shared pointwise Conv1D+BN+ReLU stack (3->64->64->64->128->1024) -> max over N -> FC head.
    The five conv layers and the max are one SharedMLPMax node (group = one cloud, K = N)."""

    # checkpoint exchange: the source holds these layers in two nn.Sequential containers, mlp_1 = [conv, bn, relu] x 2 and
    # mlp_2 = [conv, bn, relu] x 3 (pointnet_base.py:7-25); papc_amd.checkpoint maps the names
    reference_names = {"convs.0": "mlp_1.0", "bns.0": "mlp_1.1", "convs.1": "mlp_1.3", "bns.1": "mlp_1.4", "convs.2": "mlp_2.0",
                       "bns.2": "mlp_2.1", "convs.3": "mlp_2.3", "bns.3": "mlp_2.4", "convs.4": "mlp_2.6", "bns.4": "mlp_2.7"}

    def __init__(self, num_classes=10, max_points=1024):
        super().__init__()
        chans = [3, 64, 64, 64, 128, max_points]
        self.convs = nn.ModuleList([nn.Conv1d(chans[i], chans[i + 1], 1) for i in range(5)])     # :8-24
        self.bns = nn.ModuleList([nn.BatchNorm1d(chans[i + 1], eps=1e-5) for i in range(5)])
        self.fc = nn.Sequential(nn.Linear(1024, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(), nn.Dropout(p=0.7),
                                nn.Linear(256, num_classes))                                       # :26-33

    def forward(self, inputs):
        x = torch.as_tensor(inputs).float()                       # [B,3,N]
        B, _, N = x.shape
        xyz = x.transpose(1, 2)                                   # rows (b,n) read straight from the planar input
        zero = _lib.const_zeros((B, 1, 3), x.device)
        # the norms are registered layers in the source (nn.Sequential mlp_1 / mlp_2): model.eval() normalises with the running
        # statistics and leaves them untouched
        spec = StackSpec(B, N, 1, N, 0, xyz_first=True, eps=self.bns[0].eps, momentum=0.9, eval_bn=not self.training)
        ps = []
        for conv, bn in zip(self.convs, self.bns):
            ps += [conv.weight, conv.bias, bn.weight, bn.bias]
        feat = shared_mlp_max(spec, [(bn.running_mean, bn.running_var) for bn in self.bns], xyz, zero, None, None, ps)  # :42-44
        if _FUSED_HEAD and _head.plain_usable(feat, self.fc[0], self.fc[2], self.fc[5], self.training):
            spec = self.__dict__.get("_head_spec")
            if spec is None:
                spec = self.__dict__["_head_spec"] = _head.HeadSpec()
            return _head.plain_head(spec, feat, self.fc[0], self.fc[2], self.fc[4], self.fc[5], training=self.training)      # :26-33, :45 on the head kernels
        return self.fc(feat)                                      # :45


def Categorical(y, num_class=16):
    """pointnet2_basic_layers.py:7-14: class labels [B,1] (or [B]) -> one-hot float32 [B,num_class,1]."""
    y = torch.as_tensor(y).long().reshape(-1)
    # (a comparison instead of F.one_hot: that one reads min / max back to the host, which a hipGraph capture cannot contain)
    return (y.reshape(-1, 1) == torch.arange(num_class, device=y.device).reshape(1, -1)).to(torch.float32).unsqueeze(2)


class _PartSegBase(nn.Module):
    """Shared decoder half of the two part-segmentation nets (segment/pointnet2/pointnet2.py:36-51, :84-99): three
    feature-propagation levels, then conv1/bn1/relu/dropout/conv2 per point.  forward(inputs) with
    inputs = (points [B,C,N], cls_label [B,1]) -> logits [B,N,num_parts]."""

    def _head_init(self, num_parts):
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128, eps=1e-5)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_parts, 1)

    def _encode(self, l0_xyz, l0_points, start_idx, plan=(None, None, None, None)):
        raise NotImplementedError

    def sa1_first_weight(self):
        """first conv weight of the first set-abstraction level (SSG: mlp_convs, MSG: first radius branch)"""
        sa1 = self.sa1
        return (sa1.mlp_convs[0] if hasattr(sa1, "mlp_convs") else sa1.conv_blocks[0][0]).weight

    def plan_sampling(self, inputs, start_idx=None, out=None):
        """Everything of one batch that depends on the coordinates only: both sampling levels (FPS + ball queries, compact plans) and the
        3-NN searches of fp2 / fp1 -> (sa1 plan, sa2 plan, fp2 neighbours, fp1 neighbours).  Pass it to forward(plan=...); a training
        loop computes it for batch i + 1 on a side stream / graph branch while batch i trains (bench_configs.py).  ``out`` = a previous
        result for the same shapes that the kernels fill in place."""
        xyz = torch.as_tensor(inputs[0] if isinstance(inputs, (tuple, list)) else inputs)
        l0_xyz = xyz[:, :3, :] if self.normal_channel else xyz
        s = _starts(start_idx, 2)
        o = out if out is not None else (None, None, None, None)
        with torch.no_grad():
            p1 = self.sa1.sample(l0_xyz, s[0], out=o[0])
            l1_xyz = p1[0].transpose(1, 2)
            p2 = self.sa2.sample(l1_xyz, s[1], out=o[1])
            l2_xyz = p2[0].transpose(1, 2)
            n2 = self.fp2.plan(l1_xyz, l2_xyz, out=o[2])
            n1 = self.fp1.plan(l0_xyz, l1_xyz, out=o[3])
        return p1, p2, n2, n1

    def forward(self, inputs, start_idx=None, plan=None, after_encode=None):
        """``after_encode``: called once the three set-abstraction levels are enqueued, before the feature-propagation levels (a training
        loop forks the next batch's sampling branch here: the decoder's kernels are small and leave most of the chip idle)."""
        xyz = torch.as_tensor(inputs[0])
        dev = xyz.device
        cls_label = Categorical(inputs[1], self.num_classes).to(dev)                               # :28
        B, C, N = xyz.shape
        l0_points = xyz                                                                            # :32-37
        l0_xyz = xyz[:, :3, :] if self.normal_channel else xyz
        pl = plan if plan is not None else (None, None, None, None)
        (l1_xyz, l1_points), (l2_xyz, l2_points), (l3_xyz, l3_points) = self._encode(l0_xyz, l0_points, start_idx, pl)
        if after_encode is not None:
            after_encode()
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)                                 # :42
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points, planned=pl[2])                  # :43
        cls_label_one_hot = cls_label.reshape(B, self.num_classes, 1).expand(B, self.num_classes, N)   # :44
        l0_points = self.fp1(l0_xyz, l1_xyz, cat_copy([cls_label_one_hot, l0_xyz.float(), l0_points.float()], 1), l1_points,
                             planned=pl[3])                                                        # :45
        rows = l0_points.transpose(1, 2).reshape(B * N, 128)            # point-major rows (a view of fp1's buffer)
        if not rows.is_cuda:
            raise _lib.PapcError("the segmentation head needs CUDA (ROCm) tensors: there is no CPU fallback")
        # :47 relu(bn1(conv1(.))) on the fused stack; eval: the running statistics normalise (bn1 IS registered in the source) and stay untouched
        spec = StackSpec(B, N, N, 1, 128, True, eps=self.bn1.eps, momentum=0.9, pool=False, eval_bn=not self.training)
        feat = shared_mlp_max(spec, _bn_buffers([self.bn1]), None, None, None, None,
                              _stack_params([self.conv1], [self.bn1]), x_rows=rows)
        x = self.drop1(feat)                                                                        # :48
        from .linear import linear_rows
        x = linear_rows(x, self.conv2.weight, self.conv2.bias)                                       # :49 (own row kernels, no library GEMM)
        return x.view(B, N, -1)                                                                     # :50 [B,N,num_parts]


class PointNet2_SSG_Seg(_PartSegBase):
    def __init__(self, name_scope='PointNet2_SSG_Seg_', num_classes=16, num_parts=50, normal_channel=False,
                 reference_quirks=False, fp_neighbours="reference"):
        super().__init__()
        additional_channel = 3 if normal_channel else 0
        self.num_classes, self.normal_channel = num_classes, normal_channel
        q = dict(reference_quirks=reference_quirks)
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=6 + additional_channel,
                                          mlp=[64, 64, 128], group_all=False, **q)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256],
                                          group_all=False, **q)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3,
                                          mlp=[256, 512, 1024], group_all=True, **q)
        f = dict(neighbours=fp_neighbours, reference_quirks=reference_quirks)
        self.fp3 = PointNetFeaturePropagation(in_channel=1280, mlp=[256, 256], **f)
        self.fp2 = PointNetFeaturePropagation(in_channel=384, mlp=[256, 128], **f)
        self.fp1 = PointNetFeaturePropagation(in_channel=128 + 16 + 6 + additional_channel, mlp=[128, 128, 128], **f)
        self._head_init(num_parts)

    def _encode(self, l0_xyz, l0_points, start_idx, plan=(None, None, None, None)):
        s = _starts(start_idx, 2)
        l1 = self.sa1(l0_xyz, l0_points, s[0], sampled=plan[0])                                    # :38
        l2 = self.sa2(l1[0], l1[1], s[1], sampled=plan[1])                                         # :39
        l3 = self.sa3(l2[0], l2[1])                                                                # :40
        return l1, l2, l3


class PointNet2_MSG_Seg(_PartSegBase):
    def __init__(self, name_scope='PointNet2_MSG_Seg_', num_classes=16, num_parts=50, normal_channel=False,
                 reference_quirks=False, fp_neighbours="reference"):
        super().__init__()
        additional_channel = 3 if normal_channel else 0
        self.num_classes, self.normal_channel = num_classes, normal_channel
        q = dict(reference_quirks=reference_quirks)
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], 3 + additional_channel,
                                             [[32, 32, 64], [64, 64, 128], [64, 96, 128]], **q)
        self.sa2 = PointNetSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 128 + 128 + 64,
                                             [[128, 128, 256], [128, 196, 256]], **q)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=512 + 3,
                                          mlp=[256, 512, 1024], group_all=True, **q)
        f = dict(neighbours=fp_neighbours, reference_quirks=reference_quirks)
        self.fp3 = PointNetFeaturePropagation(in_channel=1536, mlp=[256, 256], **f)
        self.fp2 = PointNetFeaturePropagation(in_channel=576, mlp=[256, 128], **f)
        self.fp1 = PointNetFeaturePropagation(in_channel=150 + additional_channel, mlp=[128, 128], **f)
        self._head_init(num_parts)

    def _encode(self, l0_xyz, l0_points, start_idx, plan=(None, None, None, None)):
        s = _starts(start_idx, 2)
        l1 = self.sa1(l0_xyz, l0_points, s[0], sampled=plan[0])                                    # :86
        l2 = self.sa2(l1[0], l1[1], s[1], sampled=plan[1])                                         # :87
        l3 = self.sa3(l2[0], l2[1])                                                                # :88
        return l1, l2, l3
