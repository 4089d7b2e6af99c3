"""Which GEMM launches of the config-3 step (PointNet++ MSG segment, B=16 N=2048) decline the row-streaming kernels, and why
(PAPC_STREAM_WHY trace of stream_gemm_try; the dW side is listed from the library profiler's kernel names)."""
import os, sys, collections, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from papc_amd.models import PointNet2_MSG_Seg
    from papc_amd.synthetic import make_clouds, make_start_idx
    from papc_amd.head import softmax_cross_entropy
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, N = 16, 2048
    m = PointNet2_MSG_Seg().to(dev).train()
    x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
    cls = (torch.arange(B).reshape(B, 1) % 16).to(dev)
    tgt = torch.randint(0, 50, (B * N,), device=dev)
    st = (torch.from_numpy(make_start_idx(B, N, 3)).to(dev), torch.from_numpy(make_start_idx(B, 512, 4)).to(dev))
    softmax_cross_entropy(m((x, cls), st).reshape(B * N, 50), tgt).backward()
    torch.cuda.synchronize()
    sys.exit(0)
env = dict(os.environ, PAPC_STREAM_WHY="1")
r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
c = collections.Counter(l.strip() for l in r.stderr.splitlines() if "declined" in l)
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, k)
if r.returncode:
    print(r.stderr[-1500:])
