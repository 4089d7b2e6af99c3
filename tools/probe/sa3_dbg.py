"""Per-stage cycle breakdown (PAPC_GEMM_DBG=1) of the tiled GEMM on the group_all (SA3) shapes: M = 4096 rows."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papc_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M = 4096
for (K, N) in [(512, 1024), (1024, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev); parts = lib.papc_mlp_gemm_parts(M); st = torch.empty(parts, 2, N, device=dev)
    for _ in range(3):
        _lib.check(lib.papc_mlp_gemm_f32(0, p(x), K, None, None, None, p(w), p(b), M, K, N, p(y), p(st), None, None), "g")
    torch.cuda.synchronize()
    print("---- K=%d N=%d" % (K, N), flush=True)
    _lib.check(lib.papc_knob_set(b"PAPC_GEMM_DBG", 1), "knob")
    _lib.check(lib.papc_mlp_gemm_f32(0, p(x), K, None, None, None, p(w), p(b), M, K, N, p(y), p(st), None, None), "g")
    torch.cuda.synchronize()
    _lib.check(lib.papc_knob_set(b"PAPC_GEMM_DBG", 0), "knob")
