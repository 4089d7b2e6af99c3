"""What the driver times: bench.py's N = 1 launch structure (two alternating hipGraphs for the step, the next batch's FPS + ball-query pyramid as
a second pair of hipGraphs on a side stream, gated on the device -- and the other structures kept selectable: the pyramid as a forked branch of the
step's graph (rounds 2-4), or split over two gated side-stream graphs with the first level's farthest-point sampling two batches ahead (round 6,
measured slower) --, B = 32, N = 4096) held to the same steps launched eagerly with in-line sampling -- loss trajectory
and the flat parameter buffer after the last step -- and the row-streaming GEMM's STORE_RED flavour repeated bit-identically while a
second stream runs the sampling kernels beside it (the hazard that flavour's LATE1 ordering closes)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, name, lr, *flags, env_extra=None, expect_fail=False):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                        "--lr", lr, "--dump-trajectory", out, *flags], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    if expect_fail:
        assert r.returncode != 0, "bench.py was expected to abort: " + r.stdout[-500:]
        return None, r.stderr
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return np.load(out), line


def test_graph_replayed_forked_step_equals_eager_inline_step(tmp_path):
    """(a) lr = 0: the weights stay put, so all steps of every structure run the same forward (fresh dropout masks every step) -- the loss
    trajectories AND the last step's gradients must be BIT-identical.  (b) lr = 1e-5: the graphs must read LIVE weights; the graph-replayed
    trajectory (losses, parameters after the last step) equals the eager one bit for bit, and so do two eager runs.  (Rounds 1-5 compared against
    a noise floor here: the gather-add backward summed with float atomics.  Since round 6 it sums over the grouping's point lists in fixed
    order -- papc_point_lists_f32 -- and a training step is bit-reproducible, so any mis-ordering between the streams of a launch structure
    shows up as a non-zero difference instead of hiding under that floor.)"""
    g, line = _bench(tmp_path, "graph0", "0")
    assert int(g["graph"]) == 1 and int(g["overlap"]) == 1, "the default bench must replay captured graphs with the overlapped sampling: " + line
    assert '"launch": "hipGraph replay' in line and "a second hipGraph on the side stream" in line
    e, _ = _bench(tmp_path, "eager0", "0", "--no-graph", "--no-overlap")
    assert int(e["graph"]) == 0 and int(e["overlap"]) == 0
    f, line_f = _bench(tmp_path, "fork0", "0", "--in-graph-fork")               # rounds 2-4: the pyramid as a forked branch of the step's graph
    assert int(f["graph"]) == 1 and "fork at sa2" in line_f
    o, line_o = _bench(tmp_path, "split0", "0", "--split-pyramid")              # round 6, opt-in: two gated side graphs, FPS1 two batches ahead, three sets
    assert int(o["graph"]) == 1 and "two hipGraphs on two side streams" in line_o
    for name, r in (("side graph", g), ("in-graph fork", f), ("split pyramid", o)):
        assert r["loss"].shape == e["loss"].shape == (10,) and np.all(np.isfinite(r["loss"]))
        assert np.array_equal(r["loss"], e["loss"]), (name, r["loss"], e["loss"])
        assert len(set(r["loss"].tolist())) == 10                                 # a new dropout mask per replay
        assert np.array_equal(r["params"], r["params0"]) and np.array_equal(e["params"], e["params0"])
        gs = float(np.max(np.abs(e["grad"])))
        dg = float(np.max(np.abs(r["grad"] - e["grad"]))) / gs
        print("lr 0, %s: losses bit-identical, gradient diff %.2e of max |g|" % (name, dg))
        # round 6: the gather-add backward sums over the grouping's point lists in fixed order (no float atomics left in the step), so the last
        # step's gradient is the same BITS whatever the launch structure -- any ordering bug between the streams shows as a non-zero difference
        assert np.array_equal(r["grad"], e["grad"]), (name, dg)
    # (b)
    g1, _ = _bench(tmp_path, "graph1", "1e-5")
    e1, _ = _bench(tmp_path, "eager1", "1e-5", "--no-graph", "--no-overlap")
    e2, _ = _bench(tmp_path, "eager2", "1e-5", "--no-graph", "--no-overlap")
    rel = lambda a, b: float(np.max(np.abs(a["loss"] - b["loss"]) / np.abs(b["loss"])))
    pd = lambda a, b: float(np.max(np.abs(a["params"] - b["params"])))
    noise_l, noise_p = rel(e2, e1), pd(e2, e1)
    dl, dp = rel(g1, e1), pd(g1, e1)
    moved = float(np.max(np.abs(e1["params"] - e1["params0"])))
    print("lr 1e-5: graph vs eager loss %.2e params %.2e | eager vs eager loss %.2e params %.2e | weights moved %.2e" % (dl, dp, noise_l, noise_p, moved))
    assert moved >= 5e-5, moved
    # a step is bit-reproducible since round 6 (no float atomics): the graph-replayed structure reading LIVE weights walks the very same trajectory
    assert noise_l == 0.0 and noise_p == 0.0, (noise_l, noise_p)
    assert np.array_equal(g1["loss"], e1["loss"]) and np.array_equal(g1["params"], e1["params"]), (dl, dp)


def test_training_run_is_bit_reproducible_at_the_reference_learning_rate(tmp_path):
    """Two runs of the timed launch structure at the reference's lr = 1e-3 (PAPC/train.py:62-65) -- where rounds 1-5 saw the float atomics' last-bit
    noise grow ~30x per step until two runs differed by percents after ten steps -- end in the SAME BITS: loss trajectory, parameters, last
    gradient.  And the eager in-line structure walks the same trajectory."""
    a, _ = _bench(tmp_path, "repro_a", "1e-3")
    b, _ = _bench(tmp_path, "repro_b", "1e-3")
    e, _ = _bench(tmp_path, "repro_e", "1e-3", "--no-graph", "--no-overlap")
    assert int(a["graph"]) == 1 and int(e["graph"]) == 0
    assert float(np.max(np.abs(a["params"] - a["params0"]))) > 1e-3          # the weights really moved
    for name, r in (("second graph run", b), ("eager in-line run", e)):
        assert np.array_equal(r["loss"], a["loss"]), (name, r["loss"], a["loss"])
        assert np.array_equal(r["params"], a["params"]), name
        assert np.array_equal(r["grad"], a["grad"]), name


def test_training_run_on_the_padded_stack_is_bit_reproducible_too(tmp_path):
    """PAPC_COMPACT=0 -- the layout clouds with few padding copies (ShapeNet-like surfaces) keep: since the list builder makes a group's padding
    copies one weighted entry, the padded stack's gather-add backward runs over the point lists as well (compact.LISTS == 2), so this layout has no
    float atomics left either: two runs of the timed structure end in the same bits."""
    env = {"PAPC_COMPACT": "0"}
    a, _ = _bench(tmp_path, "pad_a", "1e-3", env_extra=env)
    b, _ = _bench(tmp_path, "pad_b", "1e-3", env_extra=env)
    assert int(a["graph"]) == 1
    assert np.array_equal(a["loss"], b["loss"]) and np.array_equal(a["params"], b["params"]) and np.array_equal(a["grad"], b["grad"])
    # ... and the float-atomic kernel (PAPC_POINT_LISTS=1: lists for compacted stacks only) walks the same trajectory to the accuracy its
    # summation-order noise allows after these few steps at this learning rate
    c, _ = _bench(tmp_path, "pad_c", "1e-3", env_extra={"PAPC_COMPACT": "0", "PAPC_POINT_LISTS": "1"})
    assert float(np.max(np.abs(c["loss"] - a["loss"]))) <= 5e-2 * float(np.max(np.abs(a["loss"])))


def test_side_graph_survives_a_main_stream_stall(tmp_path):
    """Round-5 review: the sampling graph's device-side gate gave up after ~35 ms and "started anyway", after which every later pyramid ran one
    gate early and could overwrite a plan under the previous step's backward.  Now (i) the gate counts openings, so a give-up cannot shift the
    sequence, (ii) a give-up is a sticky error bench.py aborts on -- a mis-ordered run cannot report a number --, (iii) with --side-events the side
    graph also waits for the previous step's end-of-step EVENT before it touches the shared buffers (opt-in: the cross-stream wait costs the step
    25-31 us).  Here the main stream is stalled ~60 ms ahead of every third timed step (a spinning kernel, > the old 35 ms
    bound): the loss trajectory at lr = 0 must equal the eager in-line structure's bit for bit -- with the default bound (the gate simply holds),
    and with a bound of ~2 ms plus --side-events (every stalled step's gate gives up: the events alone keep the buffers ordered; bench.py must
    report the give-ups, and abort unless told otherwise)."""
    import json
    e, _ = _bench(tmp_path, "eager0", "0", "--no-graph", "--no-overlap")
    g, line = _bench(tmp_path, "stall0", "0", "--diag-stall-ms", "60")
    assert int(g["graph"]) == 1 and "a second hipGraph on the side stream" in line
    assert json.loads(line)["config"]["gate_timeouts"] == 0
    g1, line1 = _bench(tmp_path, "stall0b", "0", "--diag-stall-ms", "60", "--split-pyramid")
    assert "two hipGraphs on two side streams" in line1 and json.loads(line1)["config"]["gate_timeouts"] == 0
    assert np.array_equal(g1["loss"], e["loss"]), (g1["loss"], e["loss"])
    assert np.array_equal(g["loss"], e["loss"]), (g["loss"], e["loss"])
    t, line_t = _bench(tmp_path, "stall1", "0", "--diag-stall-ms", "60", "--allow-gate-timeout", "--side-events", env_extra={"PAPC_GATE_SPINS": "2000"})
    assert json.loads(line_t)["config"]["gate_timeouts"] >= 2, line_t
    assert np.array_equal(t["loss"], e["loss"]), (t["loss"], e["loss"])
    gs = float(np.max(np.abs(e["grad"])))
    for name, r in (("gate holds", g), ("gate gives up", t)):
        dg = float(np.max(np.abs(r["grad"] - e["grad"]))) / gs
        print("stalled main stream, %s: losses bit-identical, gradient diff %.2e of max |g|" % (name, dg))
        assert np.array_equal(r["grad"], e["grad"]), (name, dg)
    _, err = _bench(tmp_path, "stall2", "0", "--diag-stall-ms", "60", env_extra={"PAPC_GATE_SPINS": "2000"}, expect_fail=True)
    assert "gave up" in err, err[-500:]


def test_store_red_stream_kernel_bit_identical_beside_sampling_stream(dev):
    """200 repeats of the dX launches that carry the BN-backward sums (stream_kernel<DY_*, STORE_RED>: asm prefetch ring + the
    epilogue's own y loads) while FPS / ball query run on a second stream: every repeat equals the first, bit for bit."""
    from papc_amd import functional as F
    from papc_amd.mlp import StackSpec, shared_mlp_max
    from papc_amd.synthetic import make_clouds, make_start_idx
    G, K, chans = 2048, 64, [128, 128, 256]            # SA2-shaped: DY_MAX 256 -> 128 and DY_DENSE 128 -> 128, both STORE_RED
    M = G * K
    torch.manual_seed(5)
    x = torch.randn(M, chans[0], device=dev)
    ps = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        ps += [torch.randn(cout, cin, device=dev) * (2.0 / cin) ** 0.5, torch.zeros(cout, device=dev), torch.rand(cout, device=dev) + 0.5,
               torch.randn(cout, device=dev) * 0.1]
    z = torch.zeros(1, 1, 3, device=dev)
    gout = torch.randn(G, chans[-1], device=dev)
    B, N = 32, 4096
    cloud = torch.from_numpy(np.ascontiguousarray(make_clouds(B, N, 3).transpose(0, 2, 1))).to(dev)
    st = torch.from_numpy(make_start_idx(B, N, 3)).to(dev)
    side = torch.cuda.Stream()
    first = None
    x.requires_grad_(True)
    for rep in range(200):
        if rep % 8 == 0:                                # keep the side stream busy: ~0.4 ms of FPS + ball query per launch group
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _, nx = F._fps_raw(cloud, 512, st)
                F._ball_query_raw([0.2], [32], cloud, nx)
        prm = [p.clone().requires_grad_(True) for p in ps]
        x.grad = None
        out = shared_mlp_max(StackSpec(1, M, G, K, chans[0] - 3, True), None, z, z, None, None, prm, x_rows=x)
        out.backward(gout)
        res = [x.grad] + [p.grad for p in prm]
        if first is None:
            first = [r.clone() for r in res]
        else:
            for a, b in zip(first, res):
                assert torch.equal(a, b), "repeat %d differs" % rep
    torch.cuda.synchronize()
