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
