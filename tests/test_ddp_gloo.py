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


def _model_worker(rank, world, port, q):
    """the real thing on CPU: FlatParams(PointNet2_SSG_Clas()), rank-sharded inputs, the two-bucket all-reduce of bench.py"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from papc_amd.distributed import FlatParams, init_from_env
    from papc_amd.models import PointNet2_SSG_Clas
    from papc_amd.synthetic import make_clouds, make_start_idx
    init_from_env(backend="gloo")
    torch.manual_seed(1000 + rank)                      # different initial weights per rank ...
    model = PointNet2_SSG_Clas(num_classes=16)
    flat = FlatParams(model)
    assert flat.numel == 1469520 and flat.data.numel() % 4 == 0
    flat.broadcast(0)                                   # ... identical after the broadcast
    w_sum = float(flat.data.double().sum())
    split = flat.offset_of(model.sa3)
    assert split == 80704 and flat.numel - split == 1388816          # [sa1 | sa2] head of the bucket, [sa3 | FC head] tail
    # every rank owns its own shard of clouds (bench.py: seed 1234 + rank)
    x = make_clouds(2, 256, 1234 + rank)
    st = make_start_idx(2, 256, 1234 + rank)
    # synthetic per-rank gradients written through the parameters' .grad views (what the backward kernels do)
    flat.zero_grad()
    g = torch.Generator().manual_seed(77 + rank)
    for p in model.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g))
    local = flat.grad.clone()
    scale, work = flat.allreduce_grads(split, None, async_op=True)   # tail bucket first (its gradients are final first) ...
    scale2 = flat.allreduce_grads(0, split)                          # ... then the head of the bucket
    if work is not None:
        work.wait()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = torch.allclose(flat.grad, sum(gathered)) and scale == scale2 == 1.0 / world
    ok = ok and model.fc3.weight.grad.data_ptr() >= flat.grad.data_ptr()          # still views of the flat buffer
    q.put((rank, w_sum, bool(ok), float(np.abs(x).sum()), st.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_bucket_allreduce_on_the_real_model_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                       # identical weights after the broadcast
    assert res[0][2] and res[1][2]                      # both buckets reduced: every element == the sum of the local gradients
    assert res[0][3] != res[1][3] and res[0][4] != res[1][4]   # the ranks work on different shards of clouds
