"""Elementwise error of papc_mlp_bwd_dx_f32 (DENSE dY) against float64, for the flavour selected by the environment."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from papc_amd import _lib
from papc_amd._lib import BwdDy, check, ptr, stream_ptr
lib = _lib.load()
dev = torch.device('cuda:0')
M, Co, Ci = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 128, 64
MAXMODE = len(sys.argv) > 2 and sys.argv[2] == 'max'
K = 32
g = torch.Generator(device='cpu').manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g)
dz, y, wt = r(M, Co), r(M, Co), r(Ci, Co) * 0.1
mean, invstd, gamma, beta = r(Co) * 0.1, 1 + 0.1 * r(Co).abs(), 1 + 0.1 * r(Co), 0.1 * r(Co)
scale = gamma * invstd; shift = beta - mean * scale
c1, c2 = r(Co) * 1e-3, r(Co) * 1e-3
t = {k: v.to(dev).contiguous() for k, v in dict(dz=dz, y=y, wt=wt, mean=mean, invstd=invstd, scale=scale, shift=shift, c1=c1, c2=c2).items()}
dy = BwdDy()
if MAXMODE:
    G = M // K
    gout = r(G, Co).to(dev)
    am = torch.randint(0, K, (G, Co), generator=g).to(torch.int32).to(dev)
    dzd = torch.zeros(G, K, Co, device=dev)
    dzd.scatter_(1, am.long().unsqueeze(1), gout.unsqueeze(1))
    t['dz'] = dzd.reshape(M, Co)
    dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = 1, None, gout.data_ptr(), am.data_ptr(), K
else:
    dy.dz_mode, dy.dz, dy.gout, dy.argmax, dy.K = 0, t['dz'].data_ptr(), None, None, 1
dy.y = t['y'].data_ptr()
dy.mean, dy.invstd, dy.scale, dy.shift, dy.c1, dy.c2 = (t[k].data_ptr() for k in ('mean', 'invstd', 'scale', 'shift', 'c1', 'c2'))
dx = torch.empty(M, Ci, device=dev)
check(lib.papc_mlp_bwd_dx_f32(ctypes.byref(dy), ptr(t['wt']), M, Ci, Co, ptr(dx), None, None, stream_ptr()), "dx")
torch.cuda.synchronize()
D = lambda k: t[k].double()
z = D('scale') * D('y') + D('shift')
p = torch.where(z > 0, D('dz'), torch.zeros_like(z))
xhat = (D('y') - D('mean')) * D('invstd')
dyv = D('scale') * ((p - D('c1')) - xhat * D('c2'))
ref = dyv @ D('wt').t()
# the kernel evaluates z in fp32: drop rows with a relu decision closer to 0 than fp32 can resolve
ok = (z.abs() > 1e-5).all(1)
err = ((dx.double() - ref).abs() / (dyv.abs() @ D('wt').t().abs()))[ok]
print("M=%d rows kept %d: max |err|/sum|terms| = %.2e, mean = %.2e, colsum rel err = %.2e" % (
    M, int(ok.sum()), float(err.max()), float(err.mean()),
    float(((dx.double() - ref)[ok].sum(0).abs() / ref[ok].abs().sum(0)).max())))
