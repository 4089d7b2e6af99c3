import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from papc_amd import functional as F, mlp as M_
from papc_amd.mlp import StackSpec, shared_mlp_max
from papc_amd.synthetic import make_clouds, make_start_idx
from tests.util import seeded_weights
dev = torch.device("cuda:0")
B, N, S, K = 8, 1024, 256, 32
x = make_clouds(B, N, 77)
xyz = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
st = torch.from_numpy(make_start_idx(B, N, 5)).to(dev)
_, new_xyz = F._fps_raw(xyz, S, st)
idx = F._ball_query_raw([0.25], [K], xyz, new_xyz)[0]
ws = seeded_weights([3, 64, 64, 128], 43)
sv = {}
for flag in (True, False):
    params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
    M_._XYZ1 = flag
    out = shared_mlp_max(StackSpec(B, N, S, K, 0, True), None, xyz, new_xyz, None, idx, params)
    sv[flag] = out.grad_fn.saved_tensors
L = 3
a, b = sv[True], sv[False]
ys_a, ys_b = a[7 + 4 * L: 7 + 5 * L], b[7 + 4 * L: 7 + 5 * L]
cs_a, cs_b = a[7 + 5 * L: 7 + 6 * L], b[7 + 5 * L: 7 + 6 * L]
print("cst0 diff", (cs_a[0] - cs_b[0]).abs().max(1).values, cs_b[0].abs().max(1).values)
print("y2 diff", float((ys_a[1] - ys_b[1]).abs().max()), float(ys_b[1].abs().max()))
xc = ys_a[0]; wf = a[7 + 6 * L]
# reference a1 from the row path
y1 = ys_b[0]; a1 = torch.relu(cs_b[0][2] * y1 + cs_b[0][3])
a1x = torch.relu(xc[:, :3] @ wf[:, :3].t() + wf[:, 3])
print("a1 diff", float((a1 - a1x).abs().max()), float(a1.abs().max()))
w2 = params[4].detach(); b2 = params[5].detach()
y2ref = a1 @ w2.t() + b2
print("y2 rows vs torch", float((ys_b[1] - y2ref).abs().max()), " y2 xyz vs torch", float((ys_a[1] - y2ref).abs().max()))
d = (ys_a[1] - y2ref).abs()
print("bad rows:", torch.nonzero(d.max(1).values > 1e-3)[:10].flatten().tolist(), "bad cols:", torch.nonzero(d.max(0).values > 1e-3)[:10].flatten().tolist())
r = d.max(1).values
bad = (r > 1e-3)
print("fraction bad rows %.4f" % float(bad.float().mean()))
print("first 256 rows bad map:", "".join("X" if v else "." for v in bad[:256].tolist()))
# does a bad row equal some OTHER row's correct result?
i = int(torch.nonzero(bad)[0])
cand = (y2ref - ys_a[1][i]).abs().max(1).values
print("bad row", i, "matches ref row", int(cand.argmin()), float(cand.min()))
