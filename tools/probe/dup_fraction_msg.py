"""Compacted / padded row ratio of every MSG branch of config 3 (PointNet2_MSG_Seg, B=16 N=2048) on the bench generator."""
import sys, torch
sys.path.insert(0, ".")
from papc_amd import functional as F_, compact as C
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device("cuda:0")
B, N = 16, 2048
x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)          # [B, 6?, N]
print("input", tuple(x.shape))
xyz = x[:, :3].transpose(1, 2).contiguous()
_, c1 = F_._fps_raw(xyz, 512, torch.from_numpy(make_start_idx(B, N, 3)).to(dev), 1e10)
for r, k in ((0.1, 32), (0.2, 64), (0.4, 128)):
    idx = F_._ball_query_raw([r], [k], xyz, c1)[0]
    cp = C.plan(idx)
    print("SA1 r=%.1f K=%3d rows %8d  compact/padded %.3f" % (r, k, idx.numel(), cp.fraction()))
_, c2 = F_._fps_raw(c1, 128, torch.from_numpy(make_start_idx(B, 512, 4)).to(dev), 1e10)
for r, k in ((0.4, 64), (0.8, 128)):
    idx = F_._ball_query_raw([r], [k], c1, c2)[0]
    cp = C.plan(idx)
    print("SA2 r=%.1f K=%3d rows %8d  compact/padded %.3f" % (r, k, idx.numel(), cp.fraction()))
