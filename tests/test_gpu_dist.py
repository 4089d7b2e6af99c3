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


@pytest.mark.parametrize("extra", [["--require-graph"], ["--no-graph"], ["--require-graph", "--dry-run"]])
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
    if "--dry-run" in extra:     # the N > 1 launch structure really got captured: three graphs per step, two alternating sets
        assert out["dry_run"] and out["graphs_per_step"] == 3 and out["graph_sets"] == 2 and out["capture_error"] is None, out
        assert "side stream" in out["sampling"] and out["loss"] == out["loss"]
        return
    assert ("hipGraph replay" in out["config"]["launch"]) == ("--no-graph" not in extra), out["config"]["launch"]
    assert "world 1" in out["config"]["collectives"] and "RCCL" in out["config"]["collectives"]
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert "two-stage" in out["config"]["collectives"]
