"""Run the SSG classifier's forward + backward twice from identical state and report which gradients differ (float atomics of the
gather-add backward are the only sanctioned source)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papc_amd.distributed import FlatParams
from papc_amd.head import softmax_cross_entropy
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds, make_labels, make_start_idx

dev = torch.device("cuda:0")
B, N = 32, 4096
torch.manual_seed(1234)
model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
flat = FlatParams(model)
x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev)
y = torch.from_numpy(make_labels(B, 16, 1234)).reshape(-1).to(dev)
s1 = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
s2 = torch.from_numpy(make_start_idx(B, 512, 1235)).to(dev)
grads, losses = [], []
for rep in range(3):
    model.__dict__.pop("_head_spec", None)      # same dropout masks every repeat
    flat.zero_grad()
    logits = model(x, (s1, s2))
    loss = softmax_cross_entropy(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    grads.append(flat.grad.clone())
    losses.append(float(loss))
print("losses", losses)
off = 0
for name, p in model.named_parameters():
    k = p.numel()
    a, b, c = (g[off:off + k] for g in grads)
    d1, d2 = float((a - b).abs().max()), float((a - c).abs().max())
    if d1 > 0 or d2 > 0:
        print("%-28s max|g| %.3e  diff run2 %.3e run3 %.3e  (rel %.1e)" % (name, float(a.abs().max()), d1, d2, max(d1, d2) / max(float(a.abs().max()), 1e-30)))
    off += k
print("done")
