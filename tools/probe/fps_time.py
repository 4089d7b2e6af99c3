"""FPS kernel time at the BASELINE shapes, per thread geometry (PAPC_FPS_THREADS / _SMALL knobs through papc_knob_set):
    python tools/probe/fps_time.py"""
import sys, torch
sys.path.insert(0, '.')
from papc_amd import functional as F, _lib
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device('cuda:0')
lib = _lib.load()


def run(B, N, S, knob, T):
    lib.papc_knob_set(knob.encode(), T)
    x = torch.from_numpy(make_clouds(B, N, 1)).to(dev)[:, :3].transpose(1, 2)
    st = torch.from_numpy(make_start_idx(B, N, 1)).to(dev)
    try:
        for _ in range(3): F._fps_raw(x, S, st)
    except Exception as e:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): F._fps_raw(x, S, st)
    e1.record(); torch.cuda.synchronize()
    lib.papc_knob_set(knob.encode(), 0)
    return e0.elapsed_time(e1) / 20 * 1e3


for (B, N, S, knob) in ((32, 4096, 512, "PAPC_FPS_THREADS"), (16, 2048, 512, "PAPC_FPS_THREADS"), (32, 512, 128, "PAPC_FPS_THREADS_SMALL"),
                        (16, 512, 128, "PAPC_FPS_THREADS_SMALL"), (8, 1024, 512, "PAPC_FPS_THREADS_SMALL")):
    row = []
    for T in (0, 64, 128, 256, 512, 1024):
        t = run(B, N, S, knob, T)
        row.append("T=%s %s" % (T if T else "auto", "%.1f us (%.3f us/it)" % (t, t / S) if t else "n/a"))
    print("B=%d N=%d S=%d: " % (B, N, S) + " | ".join(row))
