"""fwd+bwd kernel-time table of ONE grouped shared-MLP stack at a given shape (run under rocprofv3 by tools/probe/stack_shape.sh):
    python tools/probe/stack_shape_time.py B N S K D radius c1,c2,c3 [compact]"""
import sys, torch
sys.path.insert(0, ".")
from papc_amd import functional as F_, compact as C
from papc_amd.mlp import StackSpec, shared_mlp_max
from papc_amd.synthetic import make_clouds
B, N, S, K, D = map(int, sys.argv[1:6])
radius = float(sys.argv[6])
cs = [int(c) for c in sys.argv[7].split(",")]
dev = torch.device("cuda:0")
torch.manual_seed(0)
xyz = torch.from_numpy(make_clouds(B, N, 3)).to(dev)[:, :3].transpose(1, 2).contiguous()
_, new_xyz = F_._fps_raw(xyz, S, None, 1e10)
idx = F_._ball_query_raw([radius], [K], xyz, new_xyz)[0]
feats = torch.randn(B, N, D, device=dev, requires_grad=True) if D else None
params, bufs = [], []
cin = D + 3
for c in cs:
    params += [torch.randn(c, cin, device=dev).mul_(cin ** -0.5).requires_grad_(), torch.zeros(c, device=dev, requires_grad=True),
               torch.ones(c, device=dev, requires_grad=True), torch.zeros(c, device=dev, requires_grad=True)]
    bufs.append((torch.zeros(c, device=dev), torch.ones(c, device=dev)))
    cin = c
spec = StackSpec(B, N, S, K, D, xyz_first=False)
if len(sys.argv) > 8 and sys.argv[8] == "compact":
    spec.compact = C.plan(idx)
g = torch.randn(B * S, cs[-1], device=dev)
for _ in range(12):
    out = shared_mlp_max(spec, bufs, xyz, new_xyz, feats, idx, params)
    out.backward(g)
torch.cuda.synchronize()
