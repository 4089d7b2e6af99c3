"""Which library (aten) ops still launch kernels in one training step?  (tuning aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from papc_amd.distributed import FlatAdam, FlatParams
from papc_amd.head import softmax_cross_entropy
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds, make_labels, make_start_idx
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
B, N = 32, 4096
torch.manual_seed(1234)
model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
flat = FlatParams(model); opt = FlatAdam(flat)
x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev); y = torch.from_numpy(make_labels(B, 16, 1234)).reshape(-1).to(dev)
s1 = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev); s2 = torch.from_numpy(make_start_idx(B, 512, 1235)).to(dev)
ONE = torch.ones((), device=dev)
def step():
    flat.zero_grad()
    loss = softmax_cross_entropy(model(x, (s1, s2)), y)
    loss.backward(ONE)
    opt.step(flat.allreduce_grads())
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
seen = []
for e in evs:
    kids = [k for k in e.cpu_children if k.name.startswith("aten::")]
    if kids:  # only leaf aten ops
        continue
    if e.device_time_total > 0 or e.self_device_time_total > 0:
        st = [s for s in (e.stack or []) if "papc_amd" in s or "bench" in s or "tools/probe" in s]
        seen.append((e.name, tuple(e.input_shapes) if e.input_shapes else (), st[0] if st else "?"))
from collections import Counter
for (n, sh, st), c in Counter(seen).most_common():
    print(c, n, st)
