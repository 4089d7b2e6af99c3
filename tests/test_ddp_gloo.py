"""The N>1 path on CPU: world_size=2 over gloo -- flat parameter bucket, broadcast, ONE gradient all-reduce."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from papc_amd.distributed import FlatParams, init_from_env
    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT weights ...
    net = nn.Sequential(nn.Linear(5, 7), nn.ReLU(), nn.Linear(7, 3))
    flat = FlatParams(net)
    flat.broadcast(0)                                   # ... and are made identical
    w0 = flat.data.clone()
    flat.zero_grad()
    x = torch.full((4, 5), float(rank + 1))
    net(x).sum().backward()                             # grads land in the flat views (in-place accumulate)
    local = flat.grad.clone()
    assert local.abs().sum() > 0 and net[0].weight.grad.data_ptr() == flat.grad.data_ptr()
    scale = flat.allreduce_grads()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = torch.allclose(flat.grad, sum(gathered)) and scale == 1.0 / world
    q.put((rank, w0.tolist(), bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                       # identical weights after broadcast
    assert res[0][2] and res[1][2]                      # all-reduced gradient == sum of the local gradients
