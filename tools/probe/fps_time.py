import sys, torch, time
sys.path.insert(0, '/root/repo')
from papc_amd import functional as F
from papc_amd.synthetic import make_clouds, make_start_idx
dev = torch.device('cuda:0')
x = torch.from_numpy(make_clouds(32, 4096, 1)).to(dev).transpose(1, 2)
st = torch.from_numpy(make_start_idx(32, 4096, 1)).to(dev)
for _ in range(3): F._fps_raw(x, 512, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): F._fps_raw(x, 512, st)
e1.record(); torch.cuda.synchronize()
print("fps N=4096->512: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
