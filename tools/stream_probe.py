"""A/B timing of the row-streaming GEMM against the tiled one on the BASELINE config-2 layer shapes (tuning harness).
    python tools/stream_probe.py [reps]
Prints, per layer and kernel family, microseconds per launch and the algorithmic TB/s for PAPC_STREAM = 0 / 1."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from papc_amd import _lib  # noqa: E402
from papc_amd.mlp import StackSpec, shared_mlp_max  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
NAMES = {3: "gemm_fwd", 6: "bwd_dx", 7: "bwd_dw"}


def knob(n, v):
    _lib.check(lib.papc_knob_set(n.encode(), int(v)), "knob")


def run(G, K, chans, reps):
    M = G * K
    x = torch.randn(M, chans[0], device=dev)
    ps = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        ps += [torch.randn(cout, cin, device=dev) * (2.0 / cin) ** 0.5, torch.zeros(cout, device=dev), torch.ones(cout, device=dev),
               torch.zeros(cout, device=dev)]
    ps = [p.requires_grad_(True) for p in ps]
    spec = StackSpec(1, M, G, K, chans[0] - 3, True)
    z = torch.zeros(1, 1, 3, device=dev)
    res = {}
    for stream in (0, 1, 0, 1):
        knob("PAPC_STREAM", stream)
        for _ in range(2):
            out = shared_mlp_max(spec, None, z, z, None, None, ps, x_rows=x)
            out.backward(torch.ones_like(out))
        torch.cuda.synchronize()
        lib.papc_prof_enable(0x3FF)
        lib.papc_prof_reset()
        for _ in range(reps):
            out = shared_mlp_max(spec, None, z, z, None, None, ps, x_rows=x)
            out.backward(torch.ones_like(out))
        torch.cuda.synchronize()
        for k in NAMES:
            ms = ctypes.c_double(0)
            cnt = ctypes.c_int64(0)
            lib.papc_prof_read(k, ctypes.byref(ms), ctypes.byref(cnt))
            res.setdefault((stream, k), []).append(1e3 * ms.value / reps)
        lib.papc_prof_enable(0)
    fw = sum(4.0 * M * (a + b) for a, b in zip(chans[:-1], chans[1:]))
    dx = sum(4.0 * M * (2 * b + a + a) for a, b in list(zip(chans[:-1], chans[1:]))[1:])   # y, dz | dz_prev, y_prev
    print("stack %s  M=%d K=%d" % (chans, M, K))
    for k, byt in ((3, fw), (6, dx), (7, None)):
        a, b = min(res[(0, k)]), min(res[(1, k)])
        extra = "" if byt is None else "   %.2f -> %.2f TB/s" % (byt / a / 1e6, byt / b / 1e6)
        print("   %-9s tiled %8.1f us   stream %8.1f us%s" % (NAMES[k], a, b, extra))


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    run(16384, 32, [64, 64, 128], reps)          # SA1 layers 2-3 (layer 1 is the coordinates-only gather layer)
    run(4096, 64, [128, 128, 256], reps)         # SA2 layers 2-3
    run(16384, 32, [64, 64, 64, 128], reps)
