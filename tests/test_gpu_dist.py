"""The N > 1 path of bench.py on the one GPU there is: the driver's exact launch line with a forced 1-rank RCCL group
(PAPC_FORCE_DIST=1): init_from_env, broadcast, the two-stage backward with the tail bucket's all-reduce in flight, the
all-reduce of the head of the bucket, barrier and the rank-0 JSON line all go through RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--require-graph"], ["--no-graph"], ["--require-graph", "--dry-run"], ["--require-graph", "--eager-sampling"],
                                   ["--require-graph", "--eager-sampling", "--dry-run"]])
def test_bench_forced_one_rank_rccl(dev, extra):
    env = dict(os.environ, PAPC_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    for attempt in range(2):     # (one retry on a fresh port: a rendezvous port can be taken between the probe and the launch)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode == 0 or "address already in use" not in r.stderr.lower():
            break
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    if "--dry-run" in extra:     # the N > 1 launch structure really got captured: two graphs per step around the tail bucket's all-reduce + the gated
                                 # sampling graph on the side stream (rounds 3-4, --eager-sampling: three graphs, sampling enqueued eagerly), two alternating sets
        n_graphs = 3 if "--eager-sampling" in extra else 2
        assert out["dry_run"] and out["graphs_per_step"] == n_graphs and out["graph_sets"] == 2 and out["capture_error"] is None, out
        assert "side stream" in out["sampling"] and out["loss"] == out["loss"]
        return
    assert ("hipGraph replay" in out["config"]["launch"]) == ("--no-graph" not in extra), out["config"]["launch"]
    assert "world 1" in out["config"]["collectives"] and "RCCL" in out["config"]["collectives"]
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert "two-stage" in out["config"]["collectives"]


def test_bench_self_launch_path(dev):
    """`python bench.py --gpus N` with no RANK in the environment re-executes itself under torch.distributed.run (bench.self_launch); on
    this 1-GPU box the same path is taken with N = 1 and a forced 1-rank RCCL group (PAPC_BENCH_SELF_LAUNCH=1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PAPC_BENCH_SELF_LAUNCH="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--require-graph"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "world 1" in out["config"]["collectives"] and "RCCL" in out["config"]["collectives"] and "hipGraph replay" in out["config"]["launch"]
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["value"] > 0


def test_bench_rejects_a_rank_count_that_differs_from_gpus(dev):
    """a launcher that started 1 rank for --gpus 2 gets a clear message, not an assert"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "self-launching" in (r.stderr + r.stdout)


def test_two_ranks_on_one_gpu_stay_identical_and_match_the_averaged_gradient_step(dev, tmp_path):
    """The only hardware check of replica consistency a 1-GPU box allows (round-5 review): `bench.py --gpus 2` with BOTH ranks on cuda:0 and a
    gloo process group (RCCL refuses two ranks on one device) -- the N > 1 launch structure as it is timed: two hipGraphs per step around the
    tail bucket's all-reduce, the gated sampling graph on the side stream, the head bucket's all-reduce, eager Adam.  (i) after 2 + W + K
    optimiser steps the two ranks' flat parameter buffers are BIT-identical (each rank trains on its own shard of clouds; identical replicas
    are what data parallelism promises); (ii) they equal a single-process emulation that runs both shards' forward + backward on the same
    weights and applies Adam to the summed gradient with scale 1/2 (the reference loop, PAPC/train.py:102-116, on a global batch of 2 x 32 with
    per-GPU BatchNorm statistics) -- bit for bit (a training step has been free of float atomics since round 6: papc_point_lists_f32)."""
    import numpy as np
    import torch
    out = str(tmp_path / "two.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PAPC_DIST_BACKEND="gloo", PAPC_DEVICE_OVERRIDE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    W, K, LR = 1, 3, 1e-5
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", str(K), "--warmup", str(W), "--no-cpu-baseline", "--require-graph", "--no-dropout",
           "--lr", str(LR), "--dump-trajectory", out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "world 2, backend gloo" in line["config"]["collectives"] and "hipGraph replay of fwd+loss+bwd (2 graph(s) per step" in line["config"]["launch"]
    r0, r1 = np.load(out), np.load(out + ".rank1.npz")
    assert np.array_equal(r0["params0"], r1["params0"])
    assert np.array_equal(r0["params"], r1["params"]), "the replicas diverged: max |diff| %.3e" % float(np.abs(r0["params"] - r1["params"]).max())
    assert not np.array_equal(r0["loss"], r1["loss"])                       # ... on different shards of clouds
    moved = float(np.abs(r0["params"] - r0["params0"]).max())
    assert moved >= 3e-5, moved

    # ---- single-process emulation of the same 2 + W + K steps (bench.py's batches: seed 1234 + rank + 100003 k, step j trains on batch j mod 4; the
    # five forward + backward passes ahead of the capture consume j = 0..4 without an optimiser step)
    from papc_amd.distributed import FlatAdam, FlatParams
    from papc_amd.models import PointNet2_SSG_Clas
    from papc_amd.synthetic import make_clouds, make_labels, make_start_idx
    B, N, NB = 32, 4096, 4
    torch.manual_seed(1234)
    model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
    model.drop1.p = model.drop2.p = 0.0
    flat = FlatParams(model)
    assert np.array_equal(flat.data.cpu().numpy(), r0["params0"])
    opt = FlatAdam(flat, lr=LR, weight_decay=1e-3)

    def batch(rank, k):
        seed = 1234 + rank + 100003 * k
        return (torch.from_numpy(make_clouds(B, N, seed)).to(dev), torch.from_numpy(make_labels(B, 16, seed)).reshape(-1).to(dev),
                (torch.from_numpy(make_start_idx(B, N, seed)).to(dev), torch.from_numpy(make_start_idx(B, 512, seed + 1)).to(dev)))

    data = {(rk, k): batch(rk, k) for rk in range(2) for k in range(NB)}
    losses0 = []
    n_opt = 2 + W + K
    for j in range(5, 5 + n_opt):
        total = torch.zeros_like(flat.grad)
        for rk in range(2):
            x, y, st = data[(rk, j % NB)]
            flat.zero_grad()
            loss, _ = model(x, st, labels=y)
            loss.backward()
            total += flat.grad
            if rk == 0:
                losses0.append(float(loss))
        flat.grad.copy_(total)
        opt.step(0.5)
    torch.cuda.synchronize()
    got_l, want_l = r0["loss"].astype(np.float64), np.array(losses0[-K:])
    dl = float(np.max(np.abs(got_l - want_l) / np.abs(want_l)))
    dp = float(np.abs(flat.data.cpu().numpy() - r0["params"]).max())
    print("two ranks on one GPU vs averaged-gradient emulation: loss %.2e rel, params %.2e abs (weights moved %.2e)" % (dl, dp, moved))
    # (round 6: no float atomics left in a step, so the emulation is held to the SAME BITS -- the gloo sum of two ranks is the fp32 sum g0 + g1 the
    # emulation forms, Adam sees identical gradients)
    assert np.array_equal(got_l.astype(np.float32), want_l.astype(np.float32)), (got_l, want_l)
    assert np.array_equal(flat.data.cpu().numpy(), r0["params"]), dp
