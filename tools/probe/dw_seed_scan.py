"""Which weight seeds keep a 4-layer stack's float64 comparison away from fp32 near-ties (a pooled / ReLU decision within rounding of a tie
flips between the fp32 kernels and the float64 reference: tests/test_gpu_mlp.py::test_backward_near_ties_explain_the_seed40_excess)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papc_amd.mlp import StackSpec, shared_mlp_max
from tests import torch_ref
from tests.util import seeded_weights
dev = torch.device("cuda:0")
for chans, K in (([64, 128, 128, 256], 64), ([64, 128, 256, 64], 48)):
    for seed in range(41, 49):
        G = 1000; M = G * K
        rng = np.random.default_rng(4)
        x = torch.from_numpy(rng.normal(size=(M, chans[0])).astype(np.float32)).to(dev)
        ws = seeded_weights(chans, seed)
        ps = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        z = torch.zeros(1, 1, 3, device=dev)
        gout = torch.from_numpy(rng.normal(size=(G, chans[-1])).astype(np.float32)).to(dev)
        out = shared_mlp_max(StackSpec(1, M, G, K, chans[0] - 3, True), None, z, z, None, None, ps, x_rows=x)
        out.backward(gout)
        p64 = [p.detach().double().requires_grad_(True) for p in ps]
        ref = torch_ref.stack_max(x.double(), [tuple(p64[4 * l:4 * l + 4]) for l in range(len(chans) - 1)], K, 1e-5)
        ref.backward(gout.double())
        errs = []
        for l in range(len(chans) - 1):
            for j in (0, 2, 3):
                g, w = ps[4 * l + j].grad.double(), p64[4 * l + j].grad
                errs.append(float((g - w).abs().max() / w.abs().max()))
        print(chans, K, "seed", seed, "max err %.2e" % max(errs), " ".join("%.1e" % e for e in errs))
