"""The other single-GPU BASELINE configs behind ``bench.py --config {msg_seg,pfn,basic}`` (BASELINE.json configs[2], [4], [0]).

Same JSON contract as the headline line (bench.py): W untimed warm-up steps, K timed steps between synchronisations, value =
units per second, ``roofline`` for the dominant kernel family from library-side HIP event pairs over the timed region,
``cpu_baseline`` = the torch-CPU port of the same config (oracle/torch_cpu_reference.py) on this box's host cores.

  msg_seg  PointNet2_MSG_Seg whole model (3-radius SA1, 2-radius SA2, group-all SA3, three feature-propagation levels, per-point
           head), B=16, N=2048: fwd + CrossEntropy + bwd + Adam       /root/reference/PAPC/models/segment/pointnet2/pointnet2.py:53-98
  pfn      PillarFeatureNet, one 12000 x 100 KITTI-shaped frame: fwd + bwd + Adam   .../pointpillars/models/bones/pillars.py:43-108
  basic    PointNet_Basic_Clas, B=8, N=1024: fwd + CrossEntropy + bwd + Adam         .../classify/pointnet_base/pointnet_base.py:4-47
"""
import ctypes
import json
import os
import time

import torch

PEAK_HBM_GBS = 8000.0
PEAK_MFMA_BF16_TFLOPS = 2500.0
K_NAMES = ["fps", "ball_query", "group", "mlp_gemm_fwd", "bn_relu_max", "bwd_bn_reduce", "bwd_dx_gemm", "bwd_dw_gemm", "pfn", "misc"]


def _stack_work(M, chans, gather_D=None, pool=True):
    """algorithmic (flop, bytes) of one conv/BN/ReLU stack per family: fwd, dX, dW (same accounting as bench.algorithmic_work)"""
    f_fwd = f_dx = b_fwd = b_dx = b_dw = 0.0
    L = len(chans) - 1
    for l in range(L):
        cin, cout = chans[l], chans[l + 1]
        dense = (l < L - 1) or not pool
        f_fwd += 2.0 * M * cin * cout
        b_fwd += 4.0 * M * (cin + cout)
        dy = 4.0 * M * cout * (2 if dense else 1)
        b_dw += dy + 4.0 * M * cin
        if l > 0:
            f_dx += 2.0 * M * cin * cout
            b_dx += dy + 4.0 * M * cin * 2
        elif gather_D:
            f_dx += 2.0 * M * gather_D * cout
            b_dx += dy + 4.0 * M * gather_D
        elif gather_D is None:       # plain input rows that need a gradient (feature propagation / per-point head)
            f_dx += 2.0 * M * cin * cout
            b_dx += dy + 4.0 * M * cin
    return {3: (f_fwd, b_fwd), 6: (f_dx, b_dx), 7: (f_fwd, b_dw)}


def _sum_work(stacks):
    tot = {}
    for w in stacks:
        for k, (f, b) in w.items():
            F, Bt = tot.get(k, (0.0, 0.0))
            tot[k] = (F + f, Bt + b)
    return tot


def _prof_read(lib):
    out = {}
    for k in range(len(K_NAMES)):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.papc_prof_read(k, ctypes.byref(ms), ctypes.byref(n))
        out[k] = (ms.value, n.value)
    return out


def run(args):
    from papc_amd import _lib
    from papc_amd.distributed import FlatAdam, FlatParams
    from papc_amd.synthetic import make_clouds, make_pillars, make_start_idx
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    assert args.gpus == 1, "--config %s is a single-GPU configuration" % args.config
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.cuda.set_stream(torch.cuda.Stream())
    lib = _lib.load()
    torch.manual_seed(1234)
    from papc_amd.head import unit_gradient
    one = unit_gradient(dev)          # seeding the backward with this tensor skips the loss's multiply-by-one launch (head.py)
    if args.config == "msg_seg":
        from papc_amd.models import PointNet2_MSG_Seg
        B, N = 16, 2048
        model = PointNet2_MSG_Seg().to(dev).train()
        x = torch.from_numpy(make_clouds(B, N, 3)).to(dev)
        cls = (torch.arange(B).reshape(B, 1) % 16).to(dev)      # (resident like the points: a host array would be an H2D copy per step)
        tgt = torch.randint(0, 50, (B * N,), device=dev)
        st = (torch.from_numpy(make_start_idx(B, N, 3)).to(dev), torch.from_numpy(make_start_idx(B, 512, 4)).to(dev))
        from papc_amd.head import softmax_cross_entropy
        loss_fn = lambda plan=None, hook=None: softmax_cross_entropy(model((x, cls), st, plan=plan, after_encode=hook).reshape(B * N, 50), tgt)
        # everything that depends on the coordinates only (FPS, the multi-radius ball queries, compact plans, the 3-NN searches of the
        # feature-propagation levels) is weight-independent: the NEXT batch's runs as a second branch of this batch's graph (side stream)
        plan_fn = lambda out=None: model.plan_sampling((x, cls), st, out=out)
        units, unit, metric = B, "point-clouds/s", "point-clouds/sec (fwd+bwd) PointNet++MSG segment B=16 N=2048"
        workload = "PointNet++MSG part-segmentation whole model fwd+bwd+Adam, B=16, N=2048, 3-radius grouping (BASELINE configs[2])"
        S1, S2 = B * 512, B * 128
        work = _sum_work([_stack_work(S1 * 32, [6, 32, 32, 64], 3), _stack_work(S1 * 64, [6, 64, 64, 128], 3), _stack_work(S1 * 128, [6, 64, 96, 128], 3),
                          _stack_work(S2 * 64, [323, 128, 128, 256], 320), _stack_work(S2 * 128, [323, 128, 196, 256], 320),
                          _stack_work(B * 128, [515, 256, 512, 1024], 512),
                          _stack_work(B * 128, [1536, 256, 256], None, False), _stack_work(B * 512, [576, 256, 128], None, False),
                          _stack_work(B * N, [150, 128, 128], None, False), _stack_work(B * N, [128, 128], None, False)])
    elif args.config == "basic":
        from papc_amd.models import PointNet_Basic_Clas
        B, N = 8, 1024
        model = PointNet_Basic_Clas(num_classes=16).to(dev).train()
        x = torch.from_numpy(make_clouds(B, N, 6)).to(dev)
        tgt = torch.randint(0, 16, (B,), device=dev)
        from papc_amd.head import softmax_cross_entropy
        loss_fn = lambda: softmax_cross_entropy(model(x), tgt)
        units, unit, metric = B, "point-clouds/s", "point-clouds/sec (fwd+bwd) PointNet-Basic B=8 N=1024"
        workload = "PointNet-Basic classify fwd+bwd+Adam, B=8, N=1024 (BASELINE configs[0]: the reference's CPU-runnable plumbing case)"
        work = _sum_work([_stack_work(B * N, [3, 64, 64, 64, 128, 1024], 0)])
    else:
        from papc_amd.pillars import PillarFeatureNet
        v, n, c = make_pillars()
        # DIAGNOSTIC (the printed value is then NOT the benchmark): PAPC_BENCH_PFN_P=<pillars> runs the same five launches on the first P pillars of
        # the frame -- at a few hundred pillars what is left is the launches' fixed cost (boundaries, dependent round trips, one-workgroup tails)
        P_diag = int(os.environ.get("PAPC_BENCH_PFN_P", "0"))
        if P_diag:
            v, n, c = v[:P_diag], n[:P_diag], c[:P_diag]
        model = PillarFeatureNet(num_filters=(64,), voxel_size=(0.16, 0.16, 4), pc_range=(0, -39.68, -3, 69.12, 39.68, 1)).to(dev).train()
        # the frame comes zero-padded behind num_points, as the reference's voxeliser produces it (point_cloud_ops.py:148 zero-initialises the
        # buffers; synthetic.make_pillars masks likewise): the kernels may load the real rows only (PAPC_PFN_ZERO_PADDED=0: every row)
        model.assume_zero_padding = os.environ.get("PAPC_PFN_ZERO_PADDED", "1") != "0"
        workload_note = "rows behind num_points are zero (voxeliser contract): real rows loaded only" if model.assume_zero_padding else "all T rows loaded"
        tv, tn, tc = torch.from_numpy(v).to(dev), torch.from_numpy(n).to(dev), torch.from_numpy(c).to(dev)
        # the layer feeds PointPillarsScatter + the 2-D backbone (out of scope): its upstream gradient is a fixed [P, 64] tensor
        gout = torch.randn(v.shape[0], 64, device=dev) * 1e-3
        loss_fn = None
        units, unit, metric = 1, "frames/s", "pillar frames/sec (fwd+bwd) PillarFeatureNet 12000 pillars x 100 points"
        workload = "PointPillars PillarFeatureNet fwd+bwd+Adam, 12000 pillars x 100 points, one KITTI-shaped frame (BASELINE configs[4]); " + workload_note
        if P_diag:
            metric += " -- DIAGNOSTIC on %d pillars: not a benchmark value" % P_diag
        P, T = v.shape[0], 100
        feat = P * T * 4 * 4.0
        # three passes over the 19.2 MB of points: the Gram pass (train-mode BN statistics AND the dense part of dW from the inputs' 10x10
        # Gram matrix, csrc/pfn.hip), the apply pass (9 -> 64 layer + BN + ReLU + max, writes [P,64] out + argmax) and the sparse backward
        # pass (one argmax row per (pillar, channel); reads gout + argmax)
        work = {8: (2.0 * P * T * 9 * 64 + 2.0 * P * T * 11 * 11, 3 * feat + P * 64 * 4.0 * 4)}
    flat = FlatParams(model)
    opt = FlatAdam(flat, lr=1e-3, weight_decay=1e-3)
    # the flat gradient bucket starts as zeros and every backward here is followed by an optimiser step, so the step kernel clears it
    # (papc_adam_step_zero_f32) instead of a clear_grad launch at the head of the next step (PAPC_ZERO_IN_ADAM=0: the separate fill)
    ZERO_IN_ADAM = os.environ.get("PAPC_ZERO_IN_ADAM", "1") != "0"
    # the optimiser launch is the last node of the captured step (FlatAdam.step_dev, step count in device memory advanced by a one-thread tick:
    # on the sampling branch where the step forks one, else at its head): an eager Adam behind a replay starts 8-20 us late
    ADAM_IN_GRAPH = os.environ.get("PAPC_ADAM_IN_GRAPH", "1") != "0"
    overlap = args.config == "msg_seg" and getattr(args, "overlap", True)
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream() if overlap else None
    # the sampling pyramid of the next batch as a second hipGraph on the side stream, gated on the device (bench.py --in-graph-fork: as a forked
    # branch of the step's graph, which costs the main chain ~60 us per replay)
    # ... for THIS config measured slower (6.15 vs 5.75 ms): the MSG layers fork their radius branches onto further streams inside the step's graph, and
    # the side graph's kernels then share hardware queues with them.  Opt-in (PAPC_SIDE_GRAPH=1); the headline config takes it by default (bench.py)
    side_graph = overlap and os.environ.get("PAPC_SIDE_GRAPH") == "1" and not getattr(args, "in_graph_fork", False) and not args.no_graph
    gate = torch.zeros(4, dtype=torch.int32, device=dev) if side_graph else None      # openings, waits, give-ups (include/papc_hip.h: papc_flag_set)
    GATE_SPINS = int(os.environ.get("PAPC_GATE_SPINS", "2400000"))                    # ~2 s; a wait that gives up is an error, checked behind the timed region
    cap = {"main": False}

    def fwd_bwd(plan_in=None, plan_out=None):
        if not ZERO_IN_ADAM:
            flat.zero_grad()
        ticked = [not ADAM_IN_GRAPH]

        def update():
            if ADAM_IN_GRAPH:
                # no sampling branch took the tick: the update advances the device step count itself (its last-finishing block) -- one launch less
                opt.step_dev(flat.allreduce_grads(), zero_grad=ZERO_IN_ADAM, self_tick=not ticked[0])

        if loss_fn is None:                        # PFN: forward + backward of the layer under a given upstream gradient
            out = model(tv, tn, tc)
            out.backward(gout)
            update()
            return out
        def fork():                                # the next batch's sampling branch fills the other graph's plan buffers in place
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if not ticked[0]:
                    opt.tick()
                    ticked[0] = True
                plan_fn(plan_out)
        fork_at = getattr(args, "fork", "sa2")      # (bench.py's flag; here: "start" = with the step, anything else = behind the encoder)
        if side_graph and cap["main"]:
            def gate_open():                       # behind the encoder: the side stream's pyramid may start; the same launch ticks the optimiser's step count
                _lib.check(lib.papc_flag_set(gate.data_ptr(), 1, opt.t_dev.data_ptr() if not ticked[0] else None, _lib.stream_ptr()), "papc_flag_set")
                ticked[0] = True
            if fork_at == "start":
                gate_open()
                loss = loss_fn(plan_in)
            else:
                loss = loss_fn(plan_in, gate_open)
            loss.backward(one)
            update()
            return loss
        if plan_out is not None and fork_at == "start":
            fork()
        if plan_out is not None and fork_at != "start":
            loss = loss_fn(plan_in, fork)          # behind SA3: the decoder's kernels are small, the branch runs beside them instead of beside SA1 / SA2
        else:
            loss = loss_fn(plan_in) if plan_in is not None else loss_fn()
        loss.backward(one)
        if plan_out is not None:
            main.wait_stream(side)                 # join: the branch is part of this step
        update()
        return loss

    def step_eager():
        loss = fwd_bwd()
        if not ADAM_IN_GRAPH:
            opt.step(flat.allreduce_grads(), zero_grad=ZERO_IN_ADAM)
        return loss

    # zero_grad + forward + loss + backward captured once into a hipGraph and replayed (as in bench.py); Adam stays an eager launch
    graph = {"g": None, "loss": None, "i": 0}

    def _clone(t):
        return None if t is None else (tuple(_clone(u) for u in t) if isinstance(t, (tuple, list)) else t.clone())

    def capture():
        torch.cuda.synchronize()
        try:
            if overlap:     # two alternating graphs: each reads one set of plan buffers and fills the other on its side branch
                bufs = [_clone(plan_fn()) for _ in range(2)]
                torch.cuda.synchronize()
                gs, losses, gside = [], [], []
                for i in range(2):
                    g = torch.cuda.CUDAGraph()
                    cap["main"] = side_graph
                    with torch.cuda.graph(g, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                        losses.append(fwd_bwd(bufs[i], None if side_graph else bufs[1 - i]))
                    cap["main"] = False
                    gs.append(g)
                    if side_graph:
                        g2 = torch.cuda.CUDAGraph()
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            with torch.cuda.graph(g2, stream=side, capture_error_mode="thread_local"):
                                _lib.check(lib.papc_flag_wait(gate.data_ptr(), GATE_SPINS, _lib.stream_ptr()), "papc_flag_wait")
                                plan_fn(bufs[1 - i])
                        main.wait_stream(side)
                        gside.append(g2)
                graph["g"], graph["loss"], graph["bufs"] = gs, losses, bufs
                graph["gside"], graph["side_ev"] = (gside if side_graph else None), [None, None]
                torch.cuda.synchronize()
                return
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                graph["loss"] = [fwd_bwd()]
            graph["g"] = [g]
        except Exception as e:   # noqa: BLE001
            import sys
            print("[bench] hipGraph capture failed (%s: %s); continuing with eager launches" % (type(e).__name__, e), file=sys.stderr)
            graph["g"] = None
            torch.cuda.synchronize()

    def step():
        if graph["g"] is None:
            return step_eager()
        i = graph["i"] % len(graph["g"])
        graph["i"] += 1
        if graph.get("gside"):
            ev = graph["side_ev"][i]
            if ev is not None:
                main.wait_event(ev)                # this step's plan buffers were filled by the previous step's side graph
            graph["g"][i].replay()
            end = torch.cuda.Event()
            end.record(main)
            with torch.cuda.stream(side):
                if graph.get("prev_end") is not None:
                    side.wait_event(graph["prev_end"])     # the buffers it fills were read by the previous step to its end: ordered by events, not by the gate
                graph["gside"][i].replay()         # starts when this step has enqueued its encoder (device-side gate); fills the other buffers
                ev = torch.cuda.Event()
                ev.record(side)
            graph["side_ev"][1 - i] = ev
            graph["prev_end"] = end
        else:
            graph["g"][i].replay()
        if not ADAM_IN_GRAPH:
            opt.step(flat.allreduce_grads(), zero_grad=ZERO_IN_ADAM)
        return graph["loss"][i]

    for _ in range(max(1, args.warmup)):
        loss = step()
    torch.cuda.synchronize()
    lib.papc_prof_enable(0x3FF)
    lib.papc_prof_reset()
    from papc_amd import layers as _layers0
    _par0 = _layers0.set_branch_streams(model, False)     # (family times: one kernel on the device at a time, see below)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for _m, _v in _par0:
        _m.branch_streams = _v
    fam = _prof_read(lib)
    lib.papc_prof_enable(0)
    cand = [k for k in fam if k in work and fam[k][0] > 0]
    dominant = max(cand, key=lambda k: fam[k][0])
    if args.profile_all:
        import sys
        for k, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
            print("  %-14s %8.3f ms/step  %4d launches/step" % (K_NAMES[k], ms / 3, n // 3), file=sys.stderr)
    if not args.no_graph:
        loss = None
        capture()
        for _ in range(2):
            loss = step()
    use_graph = graph["g"] is not None
    if not use_graph:
        lib.papc_prof_enable(1 << dominant)
        lib.papc_prof_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_roof = args.steps
    if use_graph:   # kernels inside a replayed graph carry no host-visible event pairs: the family is timed on the same kernels, launched eagerly
        n_roof = min(args.steps, 20)
        lib.papc_prof_enable(1 << dominant)
        lib.papc_prof_reset()
        # kernel quality is measured with the kernels ALONE on the device: the MSG layers' parallel branch streams (layers.py) are a
        # throughput device of the timed region -- beside another branch's kernels a launch's begin-to-end time says nothing about it
        from papc_amd import layers as _layers
        _par = _layers.set_branch_streams(model, False)
        for _ in range(n_roof):
            step_eager()
        torch.cuda.synchronize()
        for _m, _v in _par:
            _m.branch_streams = _v
    dom_ms, dom_n = _prof_read(lib)[dominant]
    lib.papc_prof_enable(0)
    final_loss = float(loss.item()) if loss.dim() == 0 else float(loss.float().mean().item())
    assert final_loss == final_loss, "loss is NaN"
    flop, byts = work[dominant]
    per_step_s = dom_ms / 1e3 / n_roof
    mfma_peak = PEAK_MFMA_BF16_TFLOPS / 6.0
    t_mfma, t_hbm = flop / (mfma_peak * 1e12), byts / (PEAK_HBM_GBS * 1e9)
    if dominant != 8 and t_mfma >= t_hbm:
        ach = flop / per_step_s / 1e12
        roof = {"bound": "mfma", "kernel": K_NAMES[dominant], "achieved": round(ach, 2), "peak": round(mfma_peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / mfma_peak, 4), "traffic": None}
    else:
        ach = byts / per_step_s / 1e9
        roof = {"bound": "hbm", "kernel": K_NAMES[dominant], "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}
    roof["algorithmic"] = {"GFLOP_per_step": round(flop / 1e9, 2), "MB_per_step": round(byts / 1e6, 1), "mfma_floor_ms": round(t_mfma * 1e3, 4),
                           "hbm_floor_ms": round(t_hbm * 1e3, 4)}
    roof["launches_per_step"] = dom_n // n_roof
    roof["ms_per_step"] = round(dom_ms / n_roof, 4)
    roof["timing"] = ("HIP event pairs around every launch of the family, %d eager steps right after the graph-replayed timed region (branches in series: one kernel on the device at a time)" % n_roof) if use_graph \
        else "HIP event pairs around every launch of the family over the timed region (eager launches)"
    # HBM traffic of the family from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/refresh_profiles.sh),
    # read back from the committed per-family summary of the latest round
    import glob
    roof["traffic_note"] = "no profiles/r*_cfg_%s_pmc_family.json found" % args.config
    fams = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_cfg_%s_pmc_family.json" % args.config)))
    if fams:
        try:
            with open(fams[-1]) as fh:
                rec = json.load(fh).get(K_NAMES[dominant])
            if rec:
                lps = max(1, dom_n // n_roof)
                roof["traffic"] = round(rec["traffic_MB_per_step"] * 1e6 / lps)         # bytes per launch of the family, like `achieved`
                if roof["bound"] == "hbm":
                    roof["frac_counter_bytes"] = round(rec["traffic_MB_per_step"] * 1e6 / per_step_s / (PEAK_HBM_GBS * 1e9), 4)
                roof["traffic_note"] = ("%s: 2 x FETCH_SIZE + WRITE_SIZE of the family per step (gfx950 FETCH_SIZE correction x2), %.1f MB measured "
                                        "against %.1f MB algorithmic per step" % (os.path.join("profiles", os.path.basename(fams[-1])), rec["traffic_MB_per_step"], byts / 1e6))
        except (OSError, ValueError, KeyError) as e:
            roof["traffic_note"] = "could not read %s: %s" % (fams[-1], e)
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import torch_cpu_reference as T
        cores = min(len(os.sched_getaffinity(0)), 64)
        t, u, sample = T.time_other_config(args.config, cores)
        cpu = {"value": round(u / t, 3), "unit": unit, "cores": cores, "kind": "port",
               "sample": "torch-CPU transliteration of the reference's op decomposition (oracle/torch_cpu_reference.py): " + sample}
    out = {"metric": metric, "value": round(units * args.steps / elapsed, 2), "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": workload, "final_loss": round(final_loss, 4), "launch": (("hipGraph replay of fwd+loss+bwd+Adam (the update, which also clears the gradient bucket, is the graph's last node)" if ADAM_IN_GRAPH else "hipGraph replay of fwd+loss+bwd, eager Adam (which also clears the gradient bucket)") if ZERO_IN_ADAM else "hipGraph replay of zero_grad+fwd+loss+bwd, eager Adam") if use_graph else "eager",
                      "sampling": (("software-pipelined: batch i+1's FPS / ball queries / compact plans / 3-NN searches run as a second hipGraph on a side stream "
                                    "(no graph edge to the step's; gated on the device behind the encoder, papc_flag_set / papc_flag_wait), two alternating sets; "
                                    "every timed step computes one full set") if graph.get("gside") else
                                   ("software-pipelined: batch i+1's FPS / ball queries / compact plans / 3-NN searches run as a second branch (side stream) of "
                                    "batch i's graph, two alternating graphs; every timed step computes one full set")) if (overlap and use_graph) else "in-line",
                      "families_ms_per_step": {K_NAMES[k]: round(v[0] / 3, 4) for k, v in fam.items() if v[0] > 0}},
           "roofline": roof, "cpu_baseline": cpu}
    print(json.dumps(out))
