"""The callers of the hot path, same constructors and forward contracts as the reference models.

  PointNet2_SSG_Clas / PointNet2_MSG_Clas  <- /root/reference/PAPC/models/classify/pointnet2/pointnet2.py:6-41, :43-75
  PointNet_Basic_Clas                      <- /root/reference/PAPC/models/classify/pointnet_base/pointnet_base.py:4-47

Inputs are ``[B,3,N]`` float32 (``[B,6,N]`` with normals).  The FC head (661 776 parameters, 0.04 GFLOP) uses
plain library ops (nn.Linear / BatchNorm1d / Dropout); everything upstream runs in libpapc_hip.so.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import PointNetSetAbstraction, PointNetSetAbstractionMsg
from .mlp import StackSpec, shared_mlp_max


def _starts(start_idx, n):
    if start_idx is None:
        return [None] * n
    assert len(start_idx) == n, "start_idx must hold one [B] tensor per FPS level"
    return list(start_idx)


class PointNet2_SSG_Clas(nn.Module):
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

    def plan_sampling(self, inputs, start_idx=None):
        """The whole weight-independent sampling pyramid of one batch (FPS1, ball query 1, FPS2, ball query 2): returns
        ((new_xyz1, idx1), (new_xyz2, idx2)).  Pass it to forward(plan=...); a training loop can compute it for batch
        i+1 on a side stream while batch i trains (see bench.py)."""
        xyz = torch.as_tensor(inputs)
        if self.normal_channel:
            xyz = xyz[:, :3, :]
        s = _starts(start_idx, 2)
        with torch.no_grad():
            p1 = self.sa1.sample(xyz, s[0])
            p2 = self.sa2.sample(p1[0].transpose(1, 2), s[1])
        return p1, p2

    def forward(self, inputs, start_idx=None, plan=None, after_sa2=None):
        """inputs [B,3,N]; ``start_idx`` = optional (s1 [B], s2 [B]) FPS start indices (the source draws them at random);
        ``plan`` = optional result of :meth:`plan_sampling` for these inputs; ``after_sa2`` = optional callable invoked
        once SA2's kernels are enqueued -- from there to the end of SA3's backward only small-grid kernels run (group_all
        layer, FC head), the window in which a side stream can sample the next batch on otherwise idle CUs."""
        xyz = torch.as_tensor(inputs)
        B = xyz.shape[0]
        if self.normal_channel:
            norm, xyz = xyz[:, 3:, :], xyz[:, :3, :]
        else:
            norm = None
        s = _starts(start_idx, 2)
        pl = plan if plan is not None else (None, None)
        l1_xyz, l1_points = self.sa1(xyz, norm, s[0], sampled=pl[0])
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, s[1], sampled=pl[1])
        if after_sa2 is not None:
            after_sa2()
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.reshape(B, 1024)
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        return self.fc3(x)


class PointNet2_MSG_Clas(nn.Module):
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
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        return self.fc3(x)


class PointNet_Basic_Clas(nn.Module):
    """Shared pointwise Conv1D+BN+ReLU stack (3->64->64->64->128->1024) -> max over N -> FC head.
    The five conv layers and the max are one SharedMLPMax node (group = one cloud, K = N)."""

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
        zero = torch.zeros(B, 1, 3, device=x.device, dtype=torch.float32)
        spec = StackSpec(B, N, 1, N, 0, xyz_first=True, eps=self.bns[0].eps, momentum=0.9)
        ps = []
        for conv, bn in zip(self.convs, self.bns):
            ps += [conv.weight, conv.bias, bn.weight, bn.bias]
        feat = shared_mlp_max(spec, [(bn.running_mean, bn.running_var) for bn in self.bns], xyz, zero, None, None, ps)  # :42-44
        return self.fc(feat)                                      # :45
