"""Which library (aten) ops still launch kernels in one msg_seg training step, and from which line of this package?  (tuning aid)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from papc_amd.distributed import FlatAdam, FlatParams
from papc_amd.head import softmax_cross_entropy
from papc_amd.models import PointNet2_MSG_Seg
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
B, N = 16, 2048
torch.manual_seed(1234)
model = PointNet2_MSG_Seg().to(dev).train()
flat = FlatParams(model); opt = FlatAdam(flat)
x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
cls = (torch.arange(B).reshape(B, 1) % 16).to(dev)
tgt = torch.randint(0, 50, (B * N,), device=dev)
st = (torch.from_numpy(make_start_idx(B, N, 3)).to(dev), torch.from_numpy(make_start_idx(B, 512, 4)).to(dev))
ONE = torch.ones((), device=dev)
def step():
    flat.zero_grad()
    loss = softmax_cross_entropy(model((x, cls), st).reshape(B * N, 50), tgt)
    loss.backward(ONE)
    opt.step(flat.allreduce_grads())
for _ in range(3): step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
LAUNCH = {"copy_", "cat", "fill_", "add_", "add", "arange", "native_dropout", "sum", "native_dropout_backward", "eq", "zeros", "zero_", "clone",
          "mul", "_to_copy", "contiguous", "ones_like", "zeros_like", "new_zeros", "masked_fill", "where", "index_select", "gather", "scatter_add"}
agg = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in LAUNCH:
            where = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if "papc_amd/" in fr.filename and "_python_dispatch" not in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("papc_amd/")[-1], fr.lineno, (fr.line or "").strip()[:90])
                    break
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            agg[(name, where, str(shp))] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (name, where, shp), n in sorted(agg.items(), key=lambda kv: kv[0][1]):
    print("%2d x %-22s %s   %s" % (n, name, where, shp))
