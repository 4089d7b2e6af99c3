import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papc_amd import _lib
from papc_amd._lib import BwdDy, BwdRed, check, ptr, stream_ptr
dev = torch.device("cuda:0")
lib = _lib.load()
def knob(n, v): check(lib.papc_knob_set(n.encode(), int(v)), "knob")
knob("PAPC_STREAM_MINTILES", 1)
torch.manual_seed(0)
def run(M, C, Cin, stream, asm, reps=1):
    knob("PAPC_STREAM", stream); knob("PAPC_STREAM_ASM", asm)
    outs = []
    for r in range(reps):
        dx = torch.full((M, Cin), 7.0, device=dev)
        parts = lib.papc_mlp_gemm_parts(M)
        red = torch.full((parts, 2, Cin), 3.0, device=dev)
        dy = BwdDy()
        dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = 0, dz.data_ptr(), None, None, 1
        dy.y = y.data_ptr()
        dy.mean, dy.invstd, dy.scale, dy.shift = (cst[i].data_ptr() for i in range(4))
        dy.c1, dy.c2 = c12[0].data_ptr(), c12[1].data_ptr()
        nr = BwdRed()
        nr.y = yp.data_ptr()
        nr.mean, nr.invstd, nr.scale, nr.shift = (pc[i].data_ptr() for i in range(4))
        nr.red_partial = red.data_ptr()
        check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, Cin, C, ptr(dx), None, ctypes.byref(nr), stream_ptr()), "dx")
        torch.cuda.synchronize()
        outs.append((dx, red.sum(0)))
    return outs
for M, C, Cin in [(4096, 128, 128), (4096, 64, 64), (8192, 128, 128), (65536, 128, 128)]:
    y = torch.randn(M, C, device=dev); dz = torch.randn(M, C, device=dev); yp = torch.randn(M, Cin, device=dev)
    cst = torch.randn(4, C, device=dev); cst[1] = cst[1].abs() + 0.5
    pc = torch.randn(4, Cin, device=dev); pc[1] = pc[1].abs() + 0.5
    c12 = torch.randn(2, C, device=dev) * 0.1
    wt = torch.randn(Cin, C, device=dev) * 0.1
    ref = run(M, C, Cin, 0, 0)[0]
    for asm in (0, 1):
        for i, (dx, red) in enumerate(run(M, C, Cin, 1, asm, reps=4)):
            bad = (dx != ref[0])
            e = (dx - ref[0]).abs().max().item()
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("M %d C %d->%d asm %d rep %d: max|d| %.3e  nbad %d  bad rows %s  bad cols %s   red err %.2e" % (
                M, C, Cin, asm, i, e, int(bad.sum()), rows[:12].tolist(), cols[:12].tolist(), (red - ref[1]).abs().max().item() / ref[1].abs().max().item()))

print("---- pattern of gross errors (asm=1, M=4096, 128->128)")
M, C, Cin = 4096, 128, 128
y = torch.randn(M, C, device=dev); dz = torch.randn(M, C, device=dev); yp = torch.randn(M, Cin, device=dev)
cst = torch.randn(4, C, device=dev); cst[1] = cst[1].abs() + 0.5
pc = torch.randn(4, Cin, device=dev); pc[1] = pc[1].abs() + 0.5
c12 = torch.randn(2, C, device=dev) * 0.1
wt = torch.randn(Cin, C, device=dev) * 0.1
ref = run(M, C, Cin, 1, 0)[0]
dx, red = run(M, C, Cin, 1, 1)[0]
bad = ((dx - ref[0]).abs() > 1e-3)
print("gross-bad elements", int(bad.sum()), "of", bad.numel())
print("by row%32:", bad.reshape(-1, 32, Cin).sum((0, 2)).tolist())
print("by col:", bad.sum(0).tolist())
print("by tile (first 32):", bad.reshape(-1, 32, Cin).sum((1, 2))[:32].tolist())
print("red partial err per col (first 16):", ((red - ref[1]).abs() / ref[1].abs().max()).flatten()[:16].tolist())

print("---- what are the wrong values?")
# dY in fp64 from the definition
sc, sh, mean, invstd = (cst[i].double() for i in range(4))
z = sc * y.double() + sh
p_ = torch.where(z > 0, dz.double(), torch.zeros_like(z))
dY = sc * p_ - (sc * c12[0].double() + sc * c12[1].double() * invstd * (y.double() - mean))
idx = bad.nonzero()[:12]
for (r, c) in idx.tolist():
    terms = dY[r] * wt[c].double()            # [C] products along k
    pref = terms.reshape(8, 16).sum(1).cumsum(0)
    print("row %d col %d: got % .5f  want % .5f   prefix sums over k blocks: %s" % (r, c, dx[r, c].item(), ref[0][r, c].item(), ["%.4f" % v for v in pref.tolist()]))
