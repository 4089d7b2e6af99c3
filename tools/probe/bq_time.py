"""ball query alone (python tools/probe/bq_time.py): us per launch at the shapes of configs 2 and 3"""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from papc_amd import functional as F
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device("cuda")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, N, S, r, k) in ((32, 4096, 512, [0.2], [32]), (32, 512, 128, [0.4], [64]), (16, 2048, 512, [0.1, 0.2, 0.4], [32, 64, 128]), (16, 512, 128, [0.4, 0.8], [64, 128])):
    x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev)
    xyz = x.transpose(1, 2)
    st = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
    idx = F.farthest_point_sample(xyz, S, st)
    new_xyz = F.index_points(xyz.contiguous(), idx)
    print("B=%d N=%d S=%d r=%s k=%s: %.1f us" % (B, N, S, r, k, timeit(lambda: F._ball_query_raw(r, k, xyz, new_xyz))))
