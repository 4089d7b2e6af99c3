import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from papc_amd import _lib
from papc_amd._lib import check, ptr, stream_ptr
dev = torch.device("cuda:0")
lib = _lib.load()
M, C1, C2 = 65536, 64, 64
torch.manual_seed(0)
xc = torch.randn(M, 4, device=dev); xc[:, 3] = 0
wf = torch.randn(C1, 4, device=dev)
w2 = torch.randn(C2, C1, device=dev) * 0.2
b2 = torch.randn(C2, device=dev)
parts = lib.papc_mlp_gemm_parts(M)
ref = torch.relu(xc[:, :3].double() @ wf[:, :3].double().t() + wf[:, 3].double()) @ w2.double().t() + b2.double()
for rep in range(3):
    y = torch.full((M, C2), float("nan"), device=dev)
    stats = torch.empty(parts, 2, C2, device=dev)
    check(lib.papc_mlp_gemm_f32(6, ptr(xc), 4, None, ptr(wf), None, ptr(w2), ptr(b2), M, C1, C2, ptr(y), ptr(stats), None, stream_ptr()), "gemm xyz")
    torch.cuda.synchronize()
    d = (y.double() - ref).abs().max(1).values
    bad = d > 1e-3
    print("rep", rep, "max err", float(d.max()), "bad rows", int(bad.sum()), "nan", int(torch.isnan(y).sum()), "first bad", torch.nonzero(bad)[:8].flatten().tolist())
# the BNRELU flavour of the same kernel on the same shape, ASM on / off
x = torch.randn(M, C1, device=dev)
sc = torch.rand(C1, device=dev) + 0.5; sh = torch.randn(C1, device=dev)
ref2 = torch.relu(x.double() * sc.double() + sh.double()) @ w2.double().t() + b2.double()
for asm in (1, 0):
    lib.papc_knob_set(b"PAPC_STREAM_ASM", asm)
    y = torch.full((M, C2), float("nan"), device=dev)
    stats = torch.empty(parts, 2, C2, device=dev)
    check(lib.papc_mlp_gemm_f32(1, ptr(x), C1, None, ptr(sc), ptr(sh), ptr(w2), ptr(b2), M, C1, C2, ptr(y), ptr(stats), None, stream_ptr()), "gemm bnrelu")
    torch.cuda.synchronize()
    d = (y.double() - ref2).abs().max(1).values
    print("BNRELU asm", asm, "max err", float(d.max()), "bad rows", int((d > 1e-3).sum()))
