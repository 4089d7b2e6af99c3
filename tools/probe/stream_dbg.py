import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_gpu_stream import _run, _knob
dev = torch.device("cuda:0")
def rel(a, b):
    a = a.detach().cpu().double().numpy(); b = b.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
_knob("PAPC_STREAM_MINTILES", 1)
for G, K, chans in [(64, 64, [128, 128, 256]), (64, 64, [128, 128, 128]), (64, 64, [64, 128, 128]), (64, 64, [128, 256]), (64,64,[128,128,128,64]), (64, 32, [64, 256, 64]), (64, 32, [64, 256, 128])]:
    for asm in (1, 0):
        _knob("PAPC_STREAM_ASM", asm)
        xs, ps, g, os_ = _run(dev, G, K, chans, 3, True, True)
        xt, pt, _, ot = _run(dev, G, K, chans, 3, False, True)
        print(chans, "asm", asm, "fwd %.1e" % rel(os_, ot), " ".join("dW%d %.1e dg%d %.1e" % (l, rel(ps[4*l].grad, pt[4*l].grad), l, rel(ps[4*l+2].grad, pt[4*l+2].grad)) for l in range(len(chans)-1)), "dx %.1e" % rel(xs.grad, xt.grad))
