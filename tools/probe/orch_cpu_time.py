"""CPU-side cost of one stack forward + backward: the library's orchestration (stack.SharedMLPStack) vs the Python launch sequence
(mlp.SharedMLPMax).  Wall time of enqueueing only (no sync inside the loop) and wall time including the GPU."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from papc_amd import mlp as M_
from papc_amd.mlp import StackSpec
from papc_amd.stack import SharedMLPStack
from tests.test_gpu_cabi import _sample
from tests.util import seeded_weights
dev = torch.device("cuda:0")
for name, (B, N, S, K, r, D, chans) in {"sa1": (32, 4096, 512, 32, 0.2, 0, [3, 64, 64, 128]), "sa2": (32, 512, 128, 64, 0.4, 128, [131, 128, 128, 256])}.items():
    xyz, new_xyz, idx = _sample(dev, B, N, S, K, r, 33)
    feats = torch.randn(B, N, D, device=dev) if D else None
    ws = seeded_weights(chans, 12)
    gout = torch.randn(B * S, chans[-1], device=dev)
    for fn in (M_.SharedMLPMax, SharedMLPStack):
        params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        def it():
            spec = StackSpec(B, N, S, K, D, True)
            out = fn.apply(spec, None, xyz, new_xyz, feats, idx, None, *params)
            out.backward(gout)
        for _ in range(5): it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): it()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s %-16s enqueue %.3f ms/iter, with GPU %.3f ms/iter" % (name, fn.__name__, 1e3 * (t1 - t0) / 30, 1e3 * (t2 - t0) / 30))
        if fn is SharedMLPStack:
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
            for _ in range(10): it()
            pr.disable(); torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
