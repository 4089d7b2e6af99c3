"""Intrinsic time of the classifier head's launches (B = 32: fc1 1024->512, fc2 512->256, fc3 256->40), back to back in a graph,
against what they cost inside the training step (r04 kernel stats: forward 3 x 14 us, backward 4 x 10 us)."""
import sys, torch
sys.path.insert(0, ".")
from papc_amd import _lib
from papc_amd._lib import ptr, stream_ptr
lib = _lib.load()
dev = torch.device("cuda:0")
B = 32
rng = torch.tensor([1, 0], device=dev, dtype=torch.int64)


def timed(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cin, cout, bn in ((1024, 512, 1), (512, 256, 1), (256, 40, 0), (8, 32, 1), (8, 32, 0), (1024, 32, 1), (64, 512, 1)):
    x = torch.randn(B, cin, device=dev); w = torch.randn(cout, cin, device=dev) * 0.03; b = torch.zeros(cout, device=dev)
    g_, be = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y, out = torch.empty(B, cout, device=dev), torch.empty(B, cout, device=dev)
    mean, inv = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    keep = torch.empty(B, cout, device=dev, dtype=torch.uint8)
    def fwd():
        if bn:
            lib.papc_head_fc_f32(ptr(x), ptr(w), ptr(b), ptr(g_), ptr(be), B, cin, cout, 1, 1e-5, 0.1, 0, 0, 0, 0.4, ptr(rng), 1, 0, ptr(y), ptr(mean), ptr(inv), ptr(keep), ptr(out), stream_ptr())
        else:
            lib.papc_head_fc_f32(ptr(x), ptr(w), ptr(b), 0, 0, B, cin, cout, 0, 0.0, 0.0, 0, 0, 0, 0.0, 0, 3, 0, 0, 0, 0, 0, ptr(out), stream_ptr())
    gn = torch.randn(B, cout, device=dev)
    dy, dw, db = torch.empty(B, cout, device=dev), torch.empty(cout, cin, device=dev), torch.empty(cout, device=dev)
    dg, dbe = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    def bwd():
        lib.papc_head_bwd_f32(ptr(gn), 0, cout, ptr(out), ptr(y), ptr(mean), ptr(inv), ptr(g_), 0.4, bn, ptr(x), B, cin, cout, ptr(dy), ptr(dw), ptr(db),
                              ptr(dg) if bn else 0, ptr(dbe) if bn else 0, 0, stream_ptr())
    fwd()
    print("fc %4d -> %3d: forward %.2f us, backward (dY from g, dW, db) %.2f us per launch incl. the boundary" % (cin, cout, timed(fwd), timed(bwd)))
