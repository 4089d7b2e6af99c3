"""Does a CU mask (hipExtStreamCreateWithCUMask) survive hipGraph capture + replay?  The sampling pyramid of config 2 (its ball query fills the chip:
45 us on 256 CUs) as a graph captured on, and replayed on, a stream masked to N CUs; also launched eagerly on the same stream.
    python tools/probe/cu_mask_graph.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds, make_start_idx

dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream())
B, N = 32, 4096
model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
x = torch.from_numpy(make_clouds(B, N, 1234)).to(dev)
s1 = torch.from_numpy(make_start_idx(B, N, 1234)).to(dev)
s2 = torch.from_numpy(make_start_idx(B, 512, 1235)).to(dev)
hip = ctypes.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(n):
    if n == 0:
        return torch.cuda.Stream()
    every = max(1, ncu // n)
    words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
    for cu in range(0, ncu, every):
        words[cu // 32] |= 1 << (cu % 32)
    hs = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), ctypes.c_uint32(len(words)), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(hs.value, device=dev)


p0 = model.plan_sampling(x, (s1, s2))
buf = tuple(tuple(t.clone() for t in lvl) for lvl in p0)
torch.cuda.synchronize()
for n in (0, 128, 32, 8):
    st = masked_stream(n)
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            model.plan_sampling(x, (s1, s2), out=buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            model.plan_sampling(x, (s1, s2), out=buf)
        e1.record(st)
        e1.synchronize()
        eager = e0.elapsed_time(e1) / 10
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            model.plan_sampling(x, (s1, s2), out=buf)
        for _ in range(3):
            g.replay()
        e0.record(st)
        for _ in range(10):
            g.replay()
        e1.record(st)
        e1.synchronize()
        graph = e0.elapsed_time(e1) / 10
    print("mask %3d CUs: pyramid eager %.3f ms, replayed graph %.3f ms" % (n, eager, graph), flush=True)
