"""Which stack path PointNet-Basic takes and its per-kernel time (library profiler)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from papc_amd.models import PointNet_Basic_Clas
from papc_amd import stack, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNet_Basic_Clas(16).to(dev).train()
x = torch.randn(8, 3, 1024, device=dev)
y = m(x)
fn = y.grad_fn
seen = set()
def walk(f):
    if f is None or f in seen: return
    seen.add(f)
    if "SharedMLPStack" in type(f).__name__: print("stack node: planes =", f.planes, "nostore =", f.nostore)
    for g, _ in f.next_functions: walk(g)
walk(fn)
y.sum().backward()
torch.cuda.synchronize()
for tag in ("fwd+bwd",):
    ev0, ev1 = torch.cuda.Event(True), torch.cuda.Event(True)
    for _ in range(5): m(x).sum().backward()
    ev0.record()
    for _ in range(20): m(x).sum().backward()
    ev1.record(); torch.cuda.synchronize()
    print(tag, "eager ms", ev0.elapsed_time(ev1) / 20)
