"""Ablation of the forward GEMM on the SA1/SA2 layer shapes (PAPC_DBG: 1 skip stores, 2 skip A loads, 4 skip MFMA)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from papc_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
def run(M, cin, cout, dbg):
    os.environ["PAPC_DBG"] = str(dbg)
    x = torch.randn(M, cin, device=dev); w = torch.randn(cout, cin, device=dev); b = torch.zeros(cout, device=dev)
    sc = torch.ones(cin, device=dev); sh = torch.zeros(cin, device=dev)
    y = torch.empty(M, cout, device=dev); st = torch.empty(lib.papc_mlp_gemm_parts(M), 2, cout, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.papc_mlp_gemm_f32(1, x.data_ptr(), cin, None, sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(), M, cin, cout, y.data_ptr(), st.data_ptr(), s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for (M, cin, cout) in [(524288, 64, 64), (524288, 64, 128), (262144, 128, 128), (262144, 128, 256)]:
    r = {d: run(M, cin, cout, d) for d in (0, 1, 2, 4, 3, 7)}
    byts = M * (cin + cout) * 4
    print("M=%d %d->%d: full %.0f us (%.2f TB/s, %.1f TF) | no-store %.0f | no-Aload %.0f | no-MFMA %.0f | no-store+no-Aload %.0f | nothing %.0f" % (
        M, cin, cout, r[0], byts / r[0] / 1e6, 2.0 * M * cin * cout / r[0] / 1e6, r[1], r[2], r[4], r[3], r[7]))
