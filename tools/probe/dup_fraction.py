"""Share of ball-query padding duplicates among the rows of the SA1 / SA2 stacks of config 2 (synthetic clouds of bench.py)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from papc_amd import functional as F
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device('cuda:0')
B, N = 32, 4096
x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev).transpose(1, 2)
s1 = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
s2 = torch.from_numpy(make_start_idx(B, 512, 1235)).to(dev)
_, nx1 = F._fps_raw(x, 512, s1)
i1 = F._ball_query_raw([0.2], [32], x, nx1)[0]
_, nx2 = F._fps_raw(nx1, 128, s2)
i2 = F._ball_query_raw([0.4], [64], nx1, nx2)[0]
for name, idx in (("SA1 K=32 r=0.2", i1), ("SA2 K=64 r=0.4", i2)):
    first = idx[:, :, :1]
    dup = (idx == first)
    dup[:, :, 0] = False
    uniq = idx.shape[2] - dup.sum(2)
    print("%s: padding duplicates %.1f %% of rows; unique neighbours per group mean %.1f min %d max %d; groups that are full %.1f %%" % (
        name, 100.0 * dup.float().mean().item(), uniq.float().mean().item(), int(uniq.min()), int(uniq.max()),
        100.0 * (uniq == idx.shape[2]).float().mean().item()))
