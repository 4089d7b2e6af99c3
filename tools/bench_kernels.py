"""Per-kernel micro-benchmarks on the BASELINE config-2 shapes (tuning harness; not part of the product or tests).
    python tools/bench_kernels.py [fps|bq|mlp|all]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from papc_amd import _lib, functional as F  # noqa: E402
from papc_amd.mlp import StackSpec, shared_mlp_max  # noqa: E402
from papc_amd.synthetic import make_clouds, make_start_idx  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fps():
    B, N = 32, 4096
    x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev).transpose(1, 2)
    st = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
    for T in ("64", "128", "256", "512", "1024"):
        os.environ["PAPC_FPS_THREADS"] = T
        try:
            ms = timeit(lambda: F._fps_raw(x, 512, st))
            print("FPS SA1 B=32 N=4096 npoint=512 threads=%s: %.3f ms  (%.3f us/iter, %.0f GB/s streamed-equivalent)" % (
                T, ms, 1e3 * ms / 512, B * 512 * N * 20 / ms / 1e6))
        except Exception as e:
            print("FPS threads=%s: %s" % (T, e))
    os.environ.pop("PAPC_FPS_THREADS")
    _, nx = F._fps_raw(x, 512, st)
    x2 = nx
    st2 = torch.from_numpy(make_start_idx(B, 512, 1)).to(dev)
    for T in ("64", "128", "256", "512"):
        os.environ["PAPC_FPS_THREADS"] = T
        ms = timeit(lambda: F._fps_raw(x2, 128, st2))
        print("FPS SA2 B=32 N=512 npoint=128 threads=%s: %.3f ms (%.3f us/iter)" % (T, ms, 1e3 * ms / 128))
    os.environ.pop("PAPC_FPS_THREADS")


def bq():
    B, N = 32, 4096
    x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev).transpose(1, 2)
    st = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
    _, nx = F._fps_raw(x, 512, st)
    ms = timeit(lambda: F._ball_query_raw([0.2], [32], x, nx))
    print("ball query SA1 r=0.2 K=32: %.3f ms (%.0f GB/s test-equivalent)" % (ms, B * 512 * N * 12 / ms / 1e6))
    ms = timeit(lambda: F._ball_query_raw([0.1, 0.2, 0.4], [16, 32, 128], x, nx))
    print("ball query SA1 3 radii one scan: %.3f ms" % ms)


def mlp():
    lib = _lib.load()
    B = 32
    shapes = [("SA1", 4096, 512, 32, 0, [64, 64, 128]), ("SA2", 512, 128, 64, 128, [128, 128, 256]), ("SA3", 128, 1, 128, 256, [256, 512, 1024])]
    for name, N, S, K, D, mlp_ in shapes:
        xyz = torch.randn(B, N, 3, device=dev)
        feats = torch.randn(B, N, D, device=dev, requires_grad=True) if D else None
        if S > 1:
            _, nx = F._fps_raw(xyz, S, None)
            idx = F._ball_query_raw([10.0], [K], xyz, nx)[0]
            idx = torch.randint(0, N, (B, S, K), device=dev, dtype=torch.int32)
        else:
            nx, idx = torch.zeros(B, 1, 3, device=dev), None
        ps = []
        cin = D + 3
        for c in mlp_:
            ps += [torch.randn(c, cin, device=dev, requires_grad=True) * (2.0 / cin) ** 0.5, torch.zeros(c, device=dev, requires_grad=True),
                   torch.ones(c, device=dev, requires_grad=True), torch.zeros(c, device=dev, requires_grad=True)]
            cin = c
        ps = [p.detach().requires_grad_(True) for p in ps]
        spec = StackSpec(B, N, S, K, D, True)
        lib.papc_prof_enable(0x3FF)
        lib.papc_prof_reset()
        n = 10
        for _ in range(n):
            out = shared_mlp_max(spec, None, xyz, nx, feats, idx, ps)
            out.backward(torch.ones_like(out))
        torch.cuda.synchronize()
        M = B * S * K
        chans = [D + 3] + mlp_
        fl = sum(2.0 * M * chans[i] * chans[i + 1] for i in range(3))
        names = ["fps", "bq", "group", "gemm_fwd", "bn_relu_max", "bwd_reduce", "bwd_dx", "bwd_dw", "pfn", "misc"]
        print("%s  M=%d  fwd GEMM flop %.2f G" % (name, M, fl / 1e9))
        for k in (3, 4, 5, 6, 7, 9):
            ms = ctypes.c_double(0)
            cnt = ctypes.c_int64(0)
            lib.papc_prof_read(k, ctypes.byref(ms), ctypes.byref(cnt))
            extra = ""
            if k in (3, 7):
                extra = "  %.1f TFLOP/s" % (fl / (ms.value / n / 1e3) / 1e12)
            print("   %-12s %.3f ms/iter (%d launches)%s" % (names[k], ms.value / n, cnt.value // n, extra))
        lib.papc_prof_enable(0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("fps", "all"):
        fps()
    if what in ("bq", "all"):
        bq()
    if what in ("mlp", "all"):
        mlp()
