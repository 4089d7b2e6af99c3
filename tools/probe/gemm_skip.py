"""Timing experiment: forward GEMM (BN+ReLU operand, store+stats epilogue) M=262144 K=128 N=128 and the dX kernel with parts of the
loop disabled (PAPC_GEMM_SKIP bits: 1 mfma, 2 consume, 4 loads, 8 barrier, 16 epilogue; results are garbage, only time counts)."""
import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
from papc_amd import _lib
from papc_amd._lib import ptr, stream_ptr, check
lib = _lib.load()
dev = torch.device('cuda:0')
M, K, N = 262144, 128, 128
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.zeros(N, device=dev)
sc = torch.ones(K, device=dev); sh = torch.zeros(K, device=dev)
y = torch.empty(M, N, device=dev); parts = lib.papc_mlp_gemm_parts(M); stats = torch.empty(parts, 2, N, device=dev)
st = stream_ptr()
def run():
    check(lib.papc_mlp_gemm_f32(1, ptr(x), K, None, ptr(sc), ptr(sh), ptr(w), ptr(b), M, K, N, ptr(y), ptr(stats), None, st), "gemm")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("fwd BNRELU 262144x128x128: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
# dX with fused BN-backward sums (DENSE dz): M=262144 Cout=128 -> Cin=128
from papc_amd._lib import BwdDy, BwdRed
ycur = torch.randn(M, N, device=dev); dz = torch.randn(M, N, device=dev); yprev = torch.randn(M, K, device=dev)
cst = [torch.rand(N, device=dev) + 0.5 for _ in range(6)]; pc = [torch.rand(K, device=dev) + 0.5 for _ in range(4)]
wt = torch.randn(K, N, device=dev) * 0.1
dzp = torch.empty(M, K, device=dev); redp = torch.empty(parts, 2, K, device=dev)
dy = BwdDy(); dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = 0, dz.data_ptr(), None, None, 1
dy.y = ycur.data_ptr(); dy.mean, dy.invstd, dy.scale, dy.shift, dy.c1, dy.c2 = (c.data_ptr() for c in cst)
nr = BwdRed(); nr.y = yprev.data_ptr(); nr.mean, nr.invstd, nr.scale, nr.shift = (c.data_ptr() for c in pc); nr.red_partial = redp.data_ptr()
def run2(red):
    check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(wt), M, K, N, ptr(dzp), None, ctypes.byref(nr) if red else None, st), "dx")
for red in (False, True):
    for _ in range(3): run2(red)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run2(red)
    e1.record(); torch.cuda.synchronize()
    print("dX dense%s 262144x128x128: %.1f us" % (" + BN-reduce" if red else "", e0.elapsed_time(e1) / 20 * 1e3))
