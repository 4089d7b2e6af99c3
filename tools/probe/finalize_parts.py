"""bn_finalize / bn_bwd_finalize time against the number of partial rows they fold (the row-streaming GEMMs fill 256 of the 768 rows)."""
import sys, torch
sys.path.insert(0, ".")
from papc_amd import _lib
from papc_amd._lib import ptr, stream_ptr
lib = _lib.load()
dev = torch.device("cuda:0")
for C in (64, 128, 256):
    for parts in (768, 512, 256, 128):
        st = torch.randn(parts, 2, C, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        o = [torch.empty(C, device=dev) for _ in range(8)]
        def f():
            lib.papc_bn_finalize_f32(ptr(st), parts, 524288, C, ptr(g), ptr(b), 1e-5, 0.9, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), ptr(o[4]), ptr(o[5]), stream_ptr())
        def h():
            lib.papc_bn_bwd_finalize_f32(ptr(st), parts, 524288, C, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), 0, stream_ptr())
        res = []
        for fn in (f, h):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                for _ in range(50): fn()
            gph.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print("C=%3d parts=%3d  bn_finalize %.2f us  bn_bwd_finalize %.2f us (per launch incl. the boundary)" % (C, parts, res[0], res[1]))
