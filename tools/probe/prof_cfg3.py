import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from papc_amd.models import PointNet2_MSG_Seg
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device('cuda:0')
B, N = 16, 2048
x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
st = (torch.from_numpy(make_start_idx(B, N, 3)).to(dev), torch.from_numpy(make_start_idx(B, 512, 4)).to(dev))
m = PointNet2_MSG_Seg().to(dev); m.train()
cls = np.arange(B).reshape(B, 1) % 16
tgt = torch.randint(0, 50, (B * N,), device=dev)
for _ in range(6):
    for p in m.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(m((x, cls), st).reshape(B * N, 50), tgt).backward()
torch.cuda.synchronize()
