"""Phase-removal timing of the plane-set GEMM (csrc/smallm.hip) on the SA3 shapes: PAPC_PG_DBG bits 1 no MFMAs, 2 no fragment
loads, 4 no LDS reads, 8 no epilogue.  Results are garbage with any bit set; only the time counts.
    python tools/probe/pg_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from papc_amd import _lib, smallm  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = [("L3 fwd", 4096, 1024, 512, 1, smallm.EPI_FWD_GMAX), ("L2 fwd", 4096, 512, 256, 1, smallm.EPI_FWD), ("L1 fwd", 4096, 256, 259, 1, smallm.EPI_FWD),
          ("dX3", 4096, 512, 1024, 1, smallm.EPI_RED), ("dW3", 1024, 512, 4096, 8, smallm.EPI_STORE), ("dW2", 512, 256, 4096, 16, smallm.EPI_STORE)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if s[0] in sys.argv[1:]]
for name, R1, R2, K, split, epi in SHAPES:
    pa = torch.zeros(lib.papc_pg_planes_bytes(R1, K), dtype=torch.uint8, device=dev)
    pb = torch.zeros(lib.papc_pg_planes_bytes(R2, K), dtype=torch.uint8, device=dev)
    c = torch.empty(split, R1, R2, device=dev)
    T = (R1 + 127) // 128
    stats = torch.empty(T, 2, R2, device=dev)
    gf = torch.empty(2, T, R2, device=dev)
    gi = torch.empty(2, T, R2, device=dev, dtype=torch.int32)
    yprev = torch.randn(R1, R2, device=dev)
    cst = torch.ones(4, R2, device=dev)
    g = smallm.PgGemm()
    g.epi, g.a, g.b, g.R1, g.R2, g.K = epi, pa.data_ptr(), pb.data_ptr(), R1, R2, K
    g.c, g.ldc, g.split, g.split_stride, g.family = c.data_ptr(), R2, split, R1 * R2, 9
    g.stats, g.gmax, g.gmin, g.amax, g.amin = stats.data_ptr(), gf[0].data_ptr(), gf[1].data_ptr(), gi[0].data_ptr(), gi[1].data_ptr()
    g.y_prev, g.mean, g.invstd, g.scale, g.shift = yprev.data_ptr(), cst[0].data_ptr(), cst[1].data_ptr(), cst[2].data_ptr(), cst[3].data_ptr()
    line = []
    combos = [(nb, ns, dbg) for nb in (1, 2) for ns in (2, 3) for dbg in ((0, 1, 2, 8, 10, 9) if os.environ.get("PG_PHASES") else (0,))]
    for nb, ns, dbg in combos:
        lib.papc_knob_set(b"PAPC_PG_DBG", dbg)
        lib.papc_knob_set(b"PAPC_PG_NB", nb)
        lib.papc_knob_set(b"PAPC_PG_NS", ns)
        st = _lib.stream_ptr()
        for _ in range(5):
            lib.papc_pg_gemm_f32(ctypes.byref(g), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.papc_pg_gemm_f32(ctypes.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
        line.append("nb%d ns%d dbg%2d %6.1f us" % (nb, ns, dbg, e0.elapsed_time(e1) * 1e3 / 50))
    lib.papc_knob_set(b"PAPC_PG_DBG", 0)
    lib.papc_knob_set(b"PAPC_PG_NB", 0)
    lib.papc_knob_set(b"PAPC_PG_NS", 0)
    flop = 2.0 * R1 * R2 * K
    print("%-7s R1 %d R2 %d K %d split %d (%.2f GF): %s" % (name, R1, R2, K, split, flop / 1e9, " | ".join(line)))
