"""Gradient error yardstick for one SA stack: HIP path vs float64 autograd, next to plain torch fp32 autograd vs float64."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from papc_amd import functional as F
from papc_amd.mlp import StackSpec, shared_mlp_max
from papc_amd.synthetic import make_clouds, make_start_idx
from tests import torch_ref
from tests.util import seeded_weights
dev = torch.device('cuda:0')
B, N, S, K, D, mlp = 4, 1024, 256, 32, 0, [64, 64, 128]
x = make_clouds(B, N, 31 + N)
xyz = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
st = torch.from_numpy(make_start_idx(B, N, 1)).to(dev)
rng = np.random.default_rng(8)
_, new_xyz = F._fps_raw(xyz, S, st)
idx = F._ball_query_raw([0.3], [K], xyz, new_xyz)[0]
ws = seeded_weights([D + 3] + mlp, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
def run(dtype, hip):
    params = []
    for (w, b, g, bt) in ws:
        params += [torch.from_numpy(a).to(dev).to(dtype).requires_grad_(True) for a in (w, b, g, bt)]
    if hip:
        out = shared_mlp_max(StackSpec(B, N, S, K, D, True), None, xyz, new_xyz, None, idx, params)
    else:
        rows = torch_ref.group(xyz.to(dtype), new_xyz.to(dtype), None, idx, True).reshape(B * S * K, D + 3)
        out = torch_ref.stack_max(rows, [tuple(params[4 * l:4 * l + 4]) for l in range(len(mlp))], K, 1e-5)
    g = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(out.shape)).astype(np.float32)).to(dev).to(out.dtype)
    out.backward(g)
    return [p.grad.double() for p in params]
ref = run(torch.float64, False)
for name, gr in (("hip", run(torch.float32, True)), ("torch fp32", run(torch.float32, False))):
    errs = []
    for l in range(len(mlp)):
        for j in (0, 2, 3):
            r = ref[4 * l + j]; errs.append("%.1e" % float((gr[4 * l + j] - r).abs().max() / r.abs().max()))
    print(name, "dW/dgamma/dbeta per layer:", errs)
