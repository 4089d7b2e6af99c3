"""Which part of the step breaks hipGraph capture?  python tools/probe/graph_bisect.py  (runs variants in subprocesses)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
import torch.nn.functional as F
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.distributed import FlatParams
from papc_amd.synthetic import make_clouds, make_labels, make_start_idx
from papc_amd import functional as PF
B, N, drop, what = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0")
x = torch.from_numpy(make_clouds(B, N, 1)).to(dev); y = torch.from_numpy(make_labels(B, 16, 1)).reshape(-1).to(dev)
st = (torch.from_numpy(make_start_idx(B, N, 1)).to(dev), torch.from_numpy(make_start_idx(B, 512, 2)).to(dev))
m = PointNet2_SSG_Clas().to(dev); m.train(); m.drop1.p = m.drop2.p = drop
flat = FlatParams(m)
def fb():
    if what == "fps":
        return PF._fps_raw(x.transpose(1, 2), 512, st[0])[1].sum()
    if what == "sa1":
        return m.sa1(x, None, st[0])[1].sum()
    flat.zero_grad(); l = F.cross_entropy(m(x, st), y)
    if what == "fwd": return l
    l.backward(); return l
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fb()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): out = fb()
g.replay(); torch.cuda.synchronize(); print("ok", float(out))
''' % ROOT
for args in [("4", "1024", "0", "all"), ("32", "4096", "0", "fps"), ("32", "4096", "0", "sa1"), ("32", "4096", "0", "fwd"), ("32", "4096", "0", "all"),
             ("32", "4096", "0.4", "all"), ("4", "4096", "0", "all"), ("32", "1024", "0", "all")]:
    r = subprocess.run([sys.executable, "-c", CHILD, *args], capture_output=True, text=True)
    print(args, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], flush=True)
