"""Timings of the other BASELINE configs on one GPU (tuning harness): config 3 (MSG SA1+SA2, B=16 N=2048), config 5 (PFN
12000x100), config 0 (PointNet-Basic B=8 N=1024).   python tools/bench_configs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from papc_amd.layers import PointNetSetAbstractionMsg
from papc_amd.models import PointNet_Basic_Clas, PointNet2_MSG_Clas, PointNet2_MSG_Seg
from papc_amd.pillars import PillarFeatureNet
from papc_amd.synthetic import make_clouds, make_pillars, make_start_idx

dev = torch.device("cuda:0")


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# ---- config 3: MSG segment SA1 + SA2 (segment/pointnet2/pointnet2.py:62-63)
B, N = 16, 2048
x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
st1 = torch.from_numpy(make_start_idx(B, N, 3)).to(dev)
st2 = torch.from_numpy(make_start_idx(B, 512, 4)).to(dev)
sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], 3 + 3, [[32, 32, 64], [64, 64, 128], [64, 96, 128]]).to(dev)
sa2 = PointNetSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 128 + 128 + 64, [[128, 128, 256], [128, 196, 256]]).to(dev)
pts = torch.cat([x, x], 1)   # 6 input channels (xyz + normals slot), as the seg model feeds it


def msg_step():
    for p in list(sa1.parameters()) + list(sa2.parameters()):
        p.grad = None
    l1_xyz, l1 = sa1(x, pts, st1)
    l2_xyz, l2 = sa2(l1_xyz, l1, st2)
    l2.square().mean().backward()


print("config 3  MSG SA1+SA2 fwd+bwd, B=16 N=2048: %.3f ms  (%.0f clouds/s)" % (timeit(msg_step), B / timeit(msg_step) * 1e3))

# ---- config 3, whole model: PointNet2_MSG_Seg (encoder + 3 feature-propagation levels + per-point head)
import numpy as np
seg = PointNet2_MSG_Seg(fp_neighbours=os.environ.get("PAPC_FP_NEIGHBOURS", "reference")).to(dev)
seg.train()
cls = np.arange(B).reshape(B, 1) % 16
tgt = torch.randint(0, 50, (B * N,), device=dev)


def seg_step():
    for p in seg.parameters():
        p.grad = None
    logits = seg((x, cls), (st1, st2))
    torch.nn.functional.cross_entropy(logits.reshape(B * N, 50), tgt).backward()


t = timeit(seg_step)
print("config 3  PointNet2_MSG_Seg whole model fwd+bwd, B=16 N=2048: %.3f ms  (%.0f clouds/s)" % (t, B / t * 1e3))

# ---- config 5: PillarFeatureNet 12000 x 100 (kitti yaml num_filters [64])
v, n, c = make_pillars()
tv, tn, tc = torch.from_numpy(v).to(dev), torch.from_numpy(n).to(dev), torch.from_numpy(c).to(dev)
pfn = PillarFeatureNet(num_filters=(64,), voxel_size=(0.16, 0.16, 4), pc_range=(0, -39.68, -3, 69.12, 39.68, 1)).to(dev)
with torch.no_grad():
    t_f = timeit(lambda: pfn(tv, tn, tc), n=50)


def pfn_step():
    for p in pfn.parameters():
        p.grad = None
    pfn(tv, tn, tc).square().mean().backward()


t_fb = timeit(pfn_step, n=50)
print("config 5  PFN 12000x100: fwd %.3f ms (%.0f GB/s of the 41.5 MB algorithmic), fwd+bwd %.3f ms" % (t_f, 41.5e-3 / t_f * 1e3, t_fb))

# ---- config 5 as a frame -> BEV pipeline: voxeliser + PFN + scatter (120k-point synthetic frame)
from papc_amd.voxel import PointPillarsScatter, points_to_voxel
rng = np.random.default_rng(0)
fr = np.empty((120000, 4), np.float32)
fr[:, 0] = rng.uniform(0, 69, 120000); fr[:, 1] = rng.uniform(-39, 39, 120000); fr[:, 2] = rng.uniform(-3, 1, 120000); fr[:, 3] = rng.uniform(0, 1, 120000)
tfr = torch.from_numpy(fr).to(dev)
KW = dict(voxel_size=(0.16, 0.16, 4.0), coors_range=(0, -39.68, -3, 69.12, 39.68, 1), max_points=100, max_voxels=12000)
t_vox = timeit(lambda: points_to_voxel(tfr, padded=True, **KW), n=30)
vv, cc, nn_, cnt = points_to_voxel(tfr, padded=True, **KW)
c4 = torch.cat([torch.zeros(12000, 1, device=dev, dtype=torch.int32), cc], 1)
scat = PointPillarsScatter([1, 64, 496, 432], num_input_features=64)
with torch.no_grad():
    ff = pfn(vv, nn_.clamp(min=1), c4)
    t_sc = timeit(lambda: scat(ff, c4, 1), n=30)
print("config 5  frame->BEV: points_to_voxel(120k pts) %.3f ms, PFN fwd %.3f ms, scatter %.3f ms" % (t_vox, t_f, t_sc))

# ---- config 0: PointNet-Basic B=8 N=1024
xb = torch.from_numpy(make_clouds(8, 1024, 6)).to(dev)
pb = PointNet_Basic_Clas(num_classes=16).to(dev)


def basic_step():
    for p in pb.parameters():
        p.grad = None
    pb(xb).square().mean().backward()


print("config 0  PointNet-Basic B=8 N=1024 fwd+bwd: %.3f ms" % timeit(basic_step))
