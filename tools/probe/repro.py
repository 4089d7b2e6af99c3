"""eager train steps of the SSG classifier, loss per step with full precision: run twice, diff the output"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papc_amd.distributed import FlatAdam, FlatParams
from papc_amd.head import softmax_cross_entropy
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds, make_labels, make_start_idx
dev = torch.device("cuda:0")
B, N = 32, 4096
torch.manual_seed(1234)
model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
flat = FlatParams(model)
opt = FlatAdam(flat, lr=1e-3, weight_decay=1e-3, eps=float(os.environ.get("EPS", "1e-8")))
x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev)
y = torch.from_numpy(make_labels(B, 16, 1234)).reshape(-1).to(dev)
s1 = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
s2 = torch.from_numpy(make_start_idx(B, 512, 1235)).to(dev)
for step in range(int(os.environ.get("STEPS", "8"))):
    flat.zero_grad()
    logits = model(x, (s1, s2))
    loss = softmax_cross_entropy(logits, y)
    loss.backward()
    opt.step(1.0)
    torch.cuda.synchronize()
    print("step %d loss %.9f  |grad| %.9e  |param| %.9e" % (step, float(loss.detach()), float(flat.grad.double().norm()), float(flat.data.double().norm())))
