#!/usr/bin/env python
"""bench.py -- the reference's headline metric on MI355X:
point-clouds/sec (fwd+bwd) of PointNet++ SSG classify, B=32 per GPU, N=4096 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch: forward (FPS, ball query, fused gather + MFMA MLP
stacks, FC head), cross-entropy, backward through every layer, one flat-bucket gradient all-reduce (N>1), one Adam
step (the reference's loop, /root/reference/PAPC/train.py:102-116).  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events (library-side event pairs on the
launch stream) around the dominant kernel family over the timed region; `cpu_baseline` times the torch-CPU port of
the reference's op decomposition (oracle/torch_cpu_reference.py) on the host cores of this box.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_MFMA_BF16_TFLOPS = 2500.0  # dense bf16 (v_mfma_f32_32x32x16_bf16); an fp32 product costs SIX bf16 MFMA products here
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6.3 TB/s measured achievable)

K_NAMES = ["fps", "ball_query", "group", "mlp_gemm_fwd", "bn_relu_max", "bwd_bn_reduce", "bwd_dx_gemm", "bwd_dw_gemm", "pfn", "misc"]
K_MLP_GEMM, K_BN_RELU_MAX, K_BWD_REDUCE, K_BWD_DX, K_BWD_DW, K_FPS, K_BQ = 3, 4, 5, 6, 7, 0, 1


def ssg_layers(B, N):
    """(M rows, [channel widths], D feature channels, source points) of the three SA stacks of PointNet2_SSG_Clas for one per-GPU batch."""
    return [(B * 512 * 32, [3, 64, 64, 128], 0, B * N), (B * 128 * 64, [131, 128, 128, 256], 128, B * 512), (B * 1 * 128, [259, 256, 512, 1024], 256, B * 128)]


def algorithmic_work(B, N):
    """ALGORITHMIC work per step per family (DESIGN.md 'Measurement'): {family: (flop, bytes)}.  Bytes = every operand
    read once and every result written once per launch (fp32): what a perfect kernel moves through HBM.
      fwd  layer: read the input rows (M x Cin; the gather is counted at its logical size) + write y (M x Cout)
      dX   layer: read y_l (+ dz_l when dense) + write dz_(l-1) (M x Cin) + read y_(l-1) for the fused BN-backward sums
      dW   layer: read y_l (+ dz_l when dense) + read the input rows (M x Cin)"""
    w = {}
    f_fwd = f_dx = b_fwd = b_dx = b_dw = 0.0
    for M, ch, D, _ in ssg_layers(B, N):
        for l in range(3):
            cin, cout = ch[l], ch[l + 1]
            dense = l < 2                                   # the last layer's dz comes from the small max-pooled gradient
            f_fwd += 2.0 * M * cin * cout
            b_fwd += 4.0 * M * (cin + cout)
            dy_bytes = 4.0 * M * cout * (2 if dense else 1)
            b_dw += dy_bytes + 4.0 * M * cin
            if l > 0:
                f_dx += 2.0 * M * cin * cout
                b_dx += dy_bytes + 4.0 * M * cin * 2
            elif D:
                f_dx += 2.0 * M * D * cout                  # only the feature columns carry gradient
                b_dx += dy_bytes + 4.0 * M * D
    w[K_MLP_GEMM] = (f_fwd, b_fwd)
    w[K_BWD_DW] = (f_fwd, b_dw)
    w[K_BWD_DX] = (f_dx, b_dx)
    w[K_FPS] = (0.0, (B * 512 * N + B * 128 * 512) * 20.0)
    w[K_BQ] = (0.0, (B * 512 * N + B * 128 * 512) * 12.0)
    w[K_BN_RELU_MAX] = (0.0, sum(M * ch[3] * 4.0 for M, ch, _, _ in ssg_layers(B, N)))
    w[K_BWD_REDUCE] = (0.0, sum(2.0 * M * (ch[1] + ch[2]) * 4.0 for M, ch, _, _ in ssg_layers(B, N)))
    return w


def moved_work(B, N, plans, rows_sa2=None):
    """Bytes (and FLOP) the launches of each GEMM family REALLY process per step: `algorithmic_work` with the tensors dropped that the
    round-3/4 paths no longer move and the compacted stack priced on its device-side row count.  ``plans`` = papc_amd.stack.LAST_PLANS (the
    path the library chose per stack: csrc/sa_mlp.hip), ``rows_sa2`` = physical rows of the compacted SA2 stack (rows[0] of its plan).
      xyz1     coordinates-only first layer through its input moments: y_1 is never stored, the second layer reads the grouped centred
               coordinates (16 B / row) instead of 4 c_1 B / row, its dX is folded into four sums per channel instead of stored (xyz_fuse),
               there is no first-layer dW pass
      nostore  the max-pooled last layer never writes its [M, c_3] output; its dX / dW read the layer's INPUT instead
      lin0     gather-add first layer: the D-wide product runs on the B*N source points; the row-sum backward is a dW-family launch (over the
               compacted stack's point lists it reads the masked dz rows only, not y: round 6)
      compact  distinct neighbours only: every row count below is the physical one
      planes   few-row stacks on the planes path: operands are three bf16 planes (6 B / element), outputs fp32
    Fused epilogue operands (the BN-backward sums' second read of y_(l-1)) are counted, L2-resident weights are not."""
    fam = {K_MLP_GEMM: [0.0, 0.0], K_BWD_DX: [0.0, 0.0], K_BWD_DW: [0.0, 0.0]}
    for M, ch, D, src_pts in ssg_layers(B, N):
        fl = plans.get((M, tuple(ch[1:])), {})
        R = float(rows_sa2) if (fl.get("compact") and rows_sa2) else float(M)
        for l in range(3):
            cin, cout = ch[l], ch[l + 1]
            last, dense = l == 2, l < 2
            if fl.get("planes"):
                fam[K_MLP_GEMM][0] += 2.0 * M * cin * cout
                fam[K_MLP_GEMM][1] += 6.0 * M * cin + 6.0 * cin * cout + 4.0 * M * cout
                n_in = cin if l > 0 else D
                if n_in:
                    fam[K_BWD_DX][0] += 2.0 * M * n_in * cout
                    fam[K_BWD_DX][1] += 6.0 * M * cout + 6.0 * n_in * cout + 4.0 * M * n_in * (2 if l > 0 else 1)
                fam[K_BWD_DW][0] += 2.0 * M * cin * cout
                fam[K_BWD_DW][1] += 6.0 * M * (cin + cout) + 8.0 * cin * cout
                continue
            stored = not (last and fl.get("nostore"))
            x1 = l == 1 and fl.get("xyz1")
            in_bytes = R * 16.0 if x1 else R * cin * 4.0
            dy_bytes = (R * cout * 4.0 * (2 if dense else 1)) if stored else R * cin * 4.0     # nostore: dX / dW read the layer's input instead of y
            if l == 0 and fl.get("xyz1"):
                continue                                   # moments: no forward GEMM, no dX, no dW pass over rows
            if l == 0 and fl.get("lin0"):
                fam[K_MLP_GEMM][0] += 2.0 * src_pts * D * cout + 6.0 * R * cout
                fam[K_MLP_GEMM][1] += 4.0 * src_pts * (D + 2 * cout) + 4.0 * R * cout
                fam[K_BWD_DX][0] += 2.0 * src_pts * D * cout
                fam[K_BWD_DX][1] += 4.0 * src_pts * (D + cout)
                fam[K_BWD_DW][0] += 2.0 * src_pts * D * cout + 6.0 * R * cout
                # the row-sum backward: y and dz rows (the atomic / two-stream list kernels) -- or, over the point lists with their moments, the
                # masked dz rows alone + 20 B per list entry + P and 48 B of moments per source point (papc_lingather_bwd_pp_f32)
                one_stream = bool(fl.get("compact")) and cout in (64, 128, 256) and os.environ.get("PAPC_LG_PP", "1") != "0" \
                    and os.environ.get("PAPC_LG_LISTS", "1") != "0" and os.environ.get("PAPC_POINT_LISTS", "1") != "0"
                rows_read = (4.0 * R * cout + 20.0 * R + src_pts * (4.0 * cout + 48.0)) if one_stream else 8.0 * R * cout
                fam[K_BWD_DW][1] += rows_read + 12.0 * src_pts * cout + 4.0 * src_pts * (D + cout)
                continue
            fam[K_MLP_GEMM][0] += 2.0 * R * cin * cout
            fam[K_MLP_GEMM][1] += in_bytes + (R * cout * 4.0 if stored else 0.0)
            if l > 0:
                k_dx = cout + (cin if not stored else 0)   # A_MAXCAT: [P | A] against [W^T ; -W^T E W]
                fam[K_BWD_DX][0] += 2.0 * R * cin * k_dx
                wr = 0.0 if (x1 and fl.get("xyz_fuse")) else R * cin * 4.0
                red = in_bytes if not x1 else R * 16.0
                fam[K_BWD_DX][1] += dy_bytes + wr + (red if stored else 0.0)
            elif D:
                fam[K_BWD_DX][0] += 2.0 * R * D * cout
                fam[K_BWD_DX][1] += dy_bytes + 4.0 * R * D
            fam[K_BWD_DW][0] += 2.0 * R * cin * cout + (0.0 if stored else 2.0 * R * cin * cin)
            fam[K_BWD_DW][1] += (dy_bytes if stored else 0.0) + in_bytes
    return {k: (v[0], v[1]) for k, v in fam.items()}


def prof_read(lib):
    out = {}
    for k in range(len(K_NAMES)):
        ms = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        lib.papc_prof_read(k, ctypes.byref(ms), ctypes.byref(n))
        out[k] = (ms.value, n.value)
    return out


def self_launch(n, force_dist=False):
    """`python bench.py --gpus N` without a launcher: become the launcher -- one process per GPU under torch.distributed.run on this
    node, rendezvous on 127.0.0.1 (the container's hostname may not resolve) -- and hand the same arguments on.  Does not return."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if force_dist:
        env["PAPC_FORCE_DIST"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(max(1, n)), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU (BASELINE config: 32)")
    ap.add_argument("--npoints", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=["ssg", "msg_seg", "pfn", "basic"], default="ssg",
                    help="ssg (default) = the headline BASELINE configs[1]; the other single-GPU configs emit the same JSON (bench_configs.py)")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=True,
                    help="(default) software-pipelined sampling: batch i+1's pyramid (FPS + ball query: weight-independent, a serial "
                    "chain on 32 of the 256 CUs) runs on a side stream / graph branch beside batch i's MLP kernels")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="sample in-line at the head of every step")
    ap.add_argument("--fork", choices=["start", "sa1", "sa2", "sa3", "loss", "sa2late", "sa2end"], default="sa2", help="where the step forks the next batch's sampling branch "
                    "(sa2late / sa2end: the branch DEPENDS on the same point as sa2 but its launches are captured later -- behind SA3's forward / behind the whole "
                    "backward -- so that the main chain's continuation is the fork node's first successor in the captured graph)")
    ap.add_argument("--side-cu-mask", type=int, default=0, help="(diagnostic) create the sampling stream with a CU mask of N CUs (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--in-graph-fork", action="store_true", help="(N = 1) the next batch's sampling pyramid as a FORKED BRANCH of the step's hipGraph (rounds 2-4) "
                    "instead of the default since round 5: a SECOND hipGraph on the side stream with no graph edge to the step's -- a forked branch costs the "
                    "main chain ~60 us per replay on MI355X whatever it holds; a device-side gate (papc_flag_wait) holds the pyramid back until the step has "
                    "enqueued SA2, plain stream events order the plan buffers across steps")
    ap.add_argument("--split-pyramid", action="store_true", help="(N = 1, opt-in; measured SLOWER on MI355X in round 6: 1.62 against 1.555 ms per step, same box) the "
                    "pyramid split over two side streams -- the first level's farthest-point sampling (a serial chain of 512 argmax steps on 32 CUs, 0.32 ms) "
                    "for the batch TWO steps ahead, beside the rest of the NEXT batch's pyramid (ball queries, second level, compact plan, point lists: 0.2 ms) "
                    "-- three alternating sets of graphs and buffers.  Built on the idea that both halves would fit into the window in which the step's own "
                    "kernels are small grids (SA3, head: 0.36 ms); they do fit, and the small-grid kernels slow down by more than the pyramid's tail costs "
                    "the backward kernels it otherwise lands on")
    ap.add_argument("--eager-sampling", action="store_true", help="(N > 1) rounds 3-4: three graphs per step with the next batch's pyramid enqueued eagerly on a "
                    "high-priority side stream behind the first; default since round 5: two graphs per step (the cut where the tail bucket's all-reduce is "
                    "issued) and the pyramid as a gated hipGraph on the side stream, as at N = 1")
    ap.add_argument("--profile-all", action="store_true", help="also print per-family kernel times (stderr)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of the timed steps eagerly instead of "
                    "replaying the captured hipGraph of zero_grad + forward + loss + backward")
    ap.add_argument("--diag-fixed-plan", action="store_true", help="DIAGNOSTIC ONLY (the printed value is NOT the benchmark): reuse one "
                    "precomputed sampling plan in every step, i.e. time the MLP branch alone, to measure what the sampling overlap costs")
    ap.add_argument("--no-padded-leg", action="store_true", help="skip the second timed region with the compacted SA2 stack forced padded "
                    "(`value_padded` in the JSON line)")
    ap.add_argument("--require-graph", action="store_true", help="fail instead of falling back to eager launches when the hipGraph capture "
                    "does not succeed (a multi-GPU run must not quietly measure the slow path)")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the steps cycle through (step j trains on batch j mod this; the side "
                    "stream loads and samples batch j + 1 meanwhile): the compacted stack's row count and the sampling pyramid see different clouds")
    ap.add_argument("--min-timed-steps", type=int, default=50, help="per-step statistics (ms_median / ms_min / ms_p95, device events around every step) are taken "
                    "over at least this many steps: when --steps is smaller, further windows of --steps steps follow the contract window")
    ap.add_argument("--allow-gate-timeout", action="store_true", help="(tests) do not abort when a device-side gate wait gave up (papc_flag_wait's sticky count)")
    ap.add_argument("--side-events", action="store_true", help="the gated sampling graph ALSO waits for the previous step's end-of-step event before it is replayed "
                    "(ordering of the shared buffers by stream events as well as by the gate).  Off by default since it was measured: the cross-stream wait costs the "
                    "step 25-31 us (1.452 against 1.421 ms, same box, EMPTY side graph) -- the counting gate orders the buffers, and a gate wait that gives up is "
                    "counted and makes bench.py abort, so a mis-ordered run cannot report a number")
    ap.add_argument("--diag-empty-side", action="store_true", help="DIAGNOSTIC ONLY (not a benchmark value): the side graph holds its gate (and --side-delay-us) but NO "
                    "pyramid -- one batch, stale plans -- : what the gate's spinning wave and the two-stream launch structure cost the step by themselves")
    ap.add_argument("--side-delay-us", type=float, default=0.0, help="(diagnostic) the gated sampling graph spins for about this long behind its gate before the pyramid "
                    "starts: with --fork sa1 it places the pyramid anywhere between the end of SA1's and of SA2's forward")
    ap.add_argument("--diag-stall-ms", type=float, default=0.0, help="(tests) stall the main stream for about this long ahead of every third timed step (a spinning "
                    "one-lane kernel): the side graph's gate must hold through it, and the plan buffers stay ordered by stream events whatever the gate does")
    ap.add_argument("--no-dropout", action="store_true", help="(tests) dropout p = 0 in the classifier head, so that a single-process emulation can reproduce a multi-rank run")
    ap.add_argument("--dry-run", action="store_true", help="capture, run the warm-up steps, print the launch structure as JSON and exit")
    ap.add_argument("--lr", type=float, default=1e-3, help="Adam learning rate (train.py:62-65: 1e-3).  (tests) A training step is a "
                    "discontinuous function of the weights -- which row wins a neighbourhood max, which side of 0 a pre-activation falls -- "
                    "so the last-bit noise of the gather-add backward's float atomics grows ~30x per step at lr = 1e-3: two runs of the SAME "
                    "launch structure differ by percents after ten steps (tools/probe/repro.py).  The launch-structure test compares "
                    "trajectories at a small lr, where the weights still move every step but the noise stays at rounding level")
    ap.add_argument("--dump-trajectory", default=None, metavar="FILE.npz", help="(tests) save the loss of every timed step and the flat "
                    "parameter buffer after the last one: tests/test_gpu_bench.py holds the graph-replayed, sampling-forked step "
                    "structure to the eager in-line one")
    args = ap.parse_args()
    force1 = os.environ.get("PAPC_BENCH_SELF_LAUNCH") == "1"   # (tests, 1-GPU box) take the self-launch path with N = 1 and a forced 1-rank RCCL group
    if (args.gpus > 1 or force1) and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, force1)
    if args.config != "ssg":
        import bench_configs
        return bench_configs.run(args)

    from papc_amd import _lib
    from papc_amd.distributed import FlatAdam, FlatParams, init_from_env
    from papc_amd.head import softmax_cross_entropy
    from papc_amd.models import PointNet2_SSG_Clas
    from papc_amd.synthetic import make_clouds, make_labels, make_start_idx

    rank, world, local = init_from_env()
    if world != max(1, args.gpus):
        raise SystemExit("[bench] --gpus %d but the launcher started %d rank(s): use `python bench.py --gpus N` (self-launching) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if world > 1 and not args.no_graph:
        args.require_graph = True                 # a multi-GPU run must not quietly measure the eager fallback
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if os.environ.get("PAPC_DEVICE_OVERRIDE") is not None:      # (tests) every rank on this device: two replicas on a 1-GPU box, gloo process group
        local = int(os.environ["PAPC_DEVICE_OVERRIDE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()
    # everything runs on one non-default stream: autograd binds its AccumulateGrad nodes to the stream of the first
    # backward, and nodes bound to the legacy default stream cannot take part in a stream capture
    torch.cuda.set_stream(torch.cuda.Stream())

    B, N = args.batch, args.npoints
    torch.manual_seed(1234)                       # same initial weights on every rank (then broadcast anyway)
    model = PointNet2_SSG_Clas(num_classes=16).to(dev)
    model.train()
    if args.no_dropout:
        model.drop1.p = model.drop2.p = 0.0
    flat = FlatParams(model)
    if dist.is_initialized():
        # the only collective before the graphs are captured runs on its OWN stream: RCCL's watchdog thread polls the end events of the
        # collectives it still lists, and HIP refuses an event query on a stream that is being captured (the capture is invalidated and
        # the watchdog aborts the process).  No collective ever touches the capturing stream before the capture below -- the eager
        # passes ahead of it run without gradient exchange -- so there is nothing for the watchdog to poll there: no settle-sleep.
        comm = torch.cuda.Stream()
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            flat.broadcast(0)
        comm.synchronize()
    params0 = flat.data.detach().cpu().numpy() if args.dump_trajectory else None
    opt = FlatAdam(flat, lr=args.lr, weight_decay=1e-3)

    # each rank owns its own shard of clouds; NB distinct resident batches, step j trains on batch j mod NB (batch 0 = the one batch of rounds 1-5)
    NB = 1 if (args.diag_fixed_plan or args.diag_empty_side) else max(1, args.batches)

    def make_batch(k):
        seed = 1234 + rank + 100003 * k
        xb = torch.from_numpy(make_clouds(B, N, seed)).to(dev)
        ints = torch.stack([torch.from_numpy(make_labels(B, 16, seed)).reshape(-1), torch.from_numpy(make_start_idx(B, N, seed)),
                            torch.from_numpy(make_start_idx(B, 512, seed + 1))]).to(dev)            # int64 [3, B]: labels, FPS start indices of SA1 / SA2
        return xb, ints

    batches = [make_batch(k) for k in range(NB)]
    # the captured graphs read their inputs from two static slots (set i trains on slot i while the side stream loads slot 1 - i with the next batch)
    slots = [tuple(t.clone() for t in batches[0]) for _ in range(3)]     # (the split pyramid cycles through three)
    # the next batch reaches its slot in two hops: an eager copy into a STAGING pair on the sampling stream ahead of the gated graph (stream order
    # keeps it behind the previous replay's read of the staging pair), then a copy node INSIDE the gated graph, behind the gate -- i.e. behind the
    # end of the previous step, the slot's last reader -- so that no cross-stream event is needed to order the slot
    stage_in = tuple(t.clone() for t in batches[0])
    stage_in2 = tuple(t.clone() for t in batches[0])
    cur = {"j": 0}                                # steps enqueued so far = index of the batch the next step trains on

    def unpack(b):
        return b[0], b[1][0], b[1][1], b[1][2]    # x [B,3,N], y [B], s1 [B], s2 [B]

    def bt(j):
        return unpack(batches[j % NB])

    def load_slot(i, j):
        """slot i <- batch j (two device copies on the current stream)"""
        slots[i][0].copy_(batches[j % NB][0])
        slots[i][1].copy_(batches[j % NB][1])

    def load_stage(st, j):
        st[0].copy_(batches[j % NB][0])
        st[1].copy_(batches[j % NB][1])

    def stage_to_slot(st, i):
        """(captured into the gated graph) slot i <- the staging pair"""
        slots[i][0].copy_(st[0])
        slots[i][1].copy_(st[1])

    # Sampling pipeline: FPS / ball query depend on the batch only (not on the weights) and FPS is a serial chain that
    # occupies B=32 of the 256 CUs, so the sampling pyramid of batch i+1 is computed on a side stream while batch i's
    # MFMA kernels own the rest of the chip.  Every step still computes exactly one pyramid (for the next batch).
    main = torch.cuda.current_stream()
    # N > 1 (sampling enqueued eagerly beside the graph replays, see ext_sampling below): a high-priority stream gets its own hardware
    # queue -- on a default-priority stream the sampling kernels queue up BEHIND the graphs already enqueued (2.81 ms vs 2.46 ms per step)
    # -- whereas as a captured branch of the N = 1 graph the high priority costs 1.5 ms per step (3.89 ms vs 2.43 ms).
    side = torch.cuda.Stream(priority=-1 if (dist.is_initialized() and not args.no_graph and args.overlap and args.eager_sampling) else int(os.environ.get("PAPC_SIDE_PRIO", "0")))   # (PAPC_SIDE_PRIO: A/B)
    if args.side_cu_mask:
        # diagnostic: the sampling stream restricted to N of the chip's CUs (hipExtStreamCreateWithCUMask; every (256 / N)-th CU, so that the XCDs
        # share them evenly) -- does keeping the pyramid off most CUs give the step's persistent kernels their CUs back?
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        every = max(1, ncu // args.side_cu_mask)
        words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
        for cu in range(0, ncu, every):
            words[cu // 32] |= 1 << (cu % 32)
        hs = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), ctypes.c_uint32(len(words)), words)
        if rc != 0:
            raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
        side = torch.cuda.ExternalStream(hs.value, device=dev)
    side_graph = not args.in_graph_fork and args.fork in ("start", "sa1", "sa2", "sa3", "loss") and not dist.is_initialized() and not args.no_graph and args.overlap and not args.diag_fixed_plan
    gate = torch.zeros(4, dtype=torch.int32, device=dev) if side_graph else None      # the words papc_flag_set / papc_flag_wait share: openings, waits, give-ups
    GATE_SPINS = int(os.environ.get("PAPC_GATE_SPINS", "2400000"))                    # ~2 s: a wait that gives up is an ERROR (sync() aborts), not a late start
    stall_gate = torch.zeros(4, dtype=torch.int32, device=dev) if args.diag_stall_ms > 0 else None

    def gate_open(counter=None):
        _lib.check(lib.papc_flag_set(gate.data_ptr(), 1, counter.data_ptr() if counter is not None else None, _lib.stream_ptr()), "papc_flag_set")

    delay_gate = torch.zeros(4, dtype=torch.int32, device=dev) if args.side_delay_us > 0 else None

    def gate_wait(slot=0):
        _lib.check(lib.papc_flag_wait_slot(gate.data_ptr(), slot, GATE_SPINS, _lib.stream_ptr()), "papc_flag_wait_slot")
        if delay_gate is not None:          # (diagnostic) nobody opens this one: it spins for its bound, then gives up
            _lib.check(lib.papc_flag_wait(delay_gate.data_ptr(), int(args.side_delay_us * 1.15), _lib.stream_ptr()), "papc_flag_wait")

    split_pyr = side_graph and args.split_pyramid       # the pyramid on two side streams, FPS1 one batch further ahead (three sets of graphs / buffers)
    side2 = torch.cuda.Stream() if split_pyr else None

    from papc_amd.head import unit_gradient
    ONE = unit_gradient(dev)                    # d(loss)/d(loss), allocated once; seeding with this tensor skips the loss's multiply-by-one launch
    state = {"plan": None, "ev": None, "last_grad": None}
    # every backward here is followed by an optimiser step: the Adam kernel clears the flat gradient bucket behind its update
    # (papc_adam_step_zero_f32), so the step has no clear_grad launch at its head (PAPC_ZERO_IN_ADAM=0: the separate fill)
    ZERO_IN_ADAM = os.environ.get("PAPC_ZERO_IN_ADAM", "1") != "0"
    # one process: the optimiser launch is the LAST NODE of the step's graph (FlatAdam.step_dev: step count in device memory, advanced by a
    # one-thread tick on the sampling branch) -- an eager Adam behind the replay starts 8-20 us after the graph's last kernel.  N > 1 keeps the
    # eager launch: the gradient all-reduce sits between the backward and the update.  PAPC_ADAM_IN_GRAPH=0: eager everywhere.
    ADAM_IN_GRAPH = os.environ.get("PAPC_ADAM_IN_GRAPH", "1") != "0" and not dist.is_initialized()

    def launch_plan(j):
        """batch j's pyramid on the side stream"""
        xb, _, s1b, s2b = bt(j)
        side.wait_stream(main)                     # the inputs (and the allocator) are ordered behind the main stream
        with torch.cuda.stream(side):
            plan = model.plan_sampling(xb, (s1b, s2b))
            ev = torch.cuda.Event()
            ev.record(side)
        for lvl in plan:
            for t in lvl:
                t.record_stream(main)              # consumed by main-stream kernels: keep the blocks alive for them
        state["plan"], state["ev"], state["plan_j"] = plan, ev, j

    def step_eager(exchange=True):
        """exchange=False (N > 1, ahead of the graph capture): forward + backward only -- no collective, no optimiser step, so the
        replicas stay identical and no collective runs on the stream about to be captured"""
        j = cur["j"]
        cur["j"] = j + 1
        x, y, s1, s2 = bt(j)
        nxt = lambda: launch_plan(j + 1)          # noqa: E731
        plan = None
        if args.overlap:
            if state["plan"] is None or state.get("plan_j") != j:
                launch_plan(j)
            plan, ev = state["plan"], state["ev"]
            main.wait_event(ev)
        # the next batch's pyramid is enqueued on the side stream beside this batch's MLP kernels (--fork sa2: only once the
        # main stream reaches SA3 -- the group_all layer, the FC head and their backward are small-grid kernels)
        if not ZERO_IN_ADAM or not exchange:
            flat.zero_grad()                       # (exchange=False: no optimiser step follows to clear the bucket)
        if ADAM_IN_GRAPH and exchange:
            opt.tick()
        if args.overlap and args.fork == "start":
            nxt()
        tap = {} if use_dist else None
        loss, _ = model(x, (s1, s2), plan=plan, after_sa2=(nxt if args.overlap and args.fork == "sa2" else None), tap=tap, labels=y)
        work = None
        if use_dist and exchange:
            l2 = tap["l2_points"]
            (g_l2,) = torch.autograd.grad(loss, [l2], [ONE])
            _, work = flat.allreduce_grads(split, None, async_op=True)
            torch.autograd.backward([l2], [g_l2])
        else:
            loss.backward(ONE)
        if exchange:
            finish(work)
        return loss

    # hipGraph of the launch-bound part of the step: ~125 kernels of 3-400 us each are otherwise issued one by one from
    # Python and the GPU idles ~12 % of the step between them.  zero_grad + forward + loss + backward are captured once
    # (static buffers: the allocations made during capture live in the graph's private pool) and replayed; the gradient
    # all-reduce and the Adam kernel (whose bias correction takes the step count as a host scalar) stay eager launches.
    # With --overlap the sampling of the NEXT batch is a second branch of the same graph (forked from the capturing stream,
    # joined at the end): two graphs alternate, one reading the plan buffers the other one fills.
    use_graph = not args.no_graph
    graph_state = {"g": None, "loss": None, "i": 0, "why": None}
    # N > 1: RCCL's watchdog thread aborts the process when a capture involves a second stream while it polls the events of the
    # collectives in flight ("operation not permitted on an event last recorded in a capturing stream", ROCm 7.0 / RCCL 2.26, seen
    # with a forced 1-rank group).  There the graphs are captured on the main stream only and the NEXT batch's sampling pyramid is
    # enqueued EAGERLY on the side stream right after the graph replay (6 launches of CPU work beside a 2.4 ms graph), ordered with
    # plain stream events outside any capture; the two graphs still alternate between the two plan buffers.
    ext_sampling = dist.is_initialized() and use_graph and args.overlap
    # ... since round 5 as a hipGraph of its own on the side stream behind a device-side gate (the N = 1 structure): one graph boundary and ten eager
    # launches less per step (--eager-sampling restores the three-graph form)
    dist_side_graph = ext_sampling and not args.eager_sampling and args.fork == "sa2" and not args.diag_fixed_plan
    if dist_side_graph and gate is None:
        gate = torch.zeros(4, dtype=torch.int32, device=dev)

    use_dist = dist.is_initialized()
    # N > 1: the backward runs in two stages around l2_points (the tensor SA3 consumes).  Stage 1 = FC head + SA3, whose
    # gradients are the tail [split, end) of the flat bucket (1 388 816 of 1 469 520 floats); their all-reduce is issued as soon
    # as stage 1 is enqueued and runs over xGMI while stage 2 (SA2 + SA1 backward, ~1.4 ms) computes; only the 80 704-float head
    # of the bucket is reduced after the last kernel.
    split = flat.offset_of(model.sa3) if use_dist else 0
    if use_dist:
        # the two-stage backward takes stage 1's parameter gradients from the kernels' in-place accumulation into the flat bucket
        # (torch.autograd.grad returns only d loss / d l2_points): every parameter must have opted in, and the fused head must be usable
        assert all(getattr(p, "_papc_inplace_grad", False) for p in model.parameters()), "two-stage backward needs FlatParams-owned parameters"
        assert 2 <= B <= 256, "two-stage backward needs the fused classifier head (2 <= B <= 256)"

    def stage1(plan_in=None, plan_out=None, cut=None, si=0):
        """``si`` = the input slot this graph trains on; a forked sampling branch reads the OTHER slot (the next batch)"""
        x, y, s1, s2 = unpack(slots[si])
        xn, _, s1n, s2n = unpack(slots[1 - si] if si < 2 else slots[0])
        if not ZERO_IN_ADAM:
            flat.zero_grad()

        ticked = [not ADAM_IN_GRAPH]

        fork_ev = [None]

        def fork(ev=None):
            if ev is None:
                side.wait_stream(main)
            else:
                side.wait_event(ev)
            with torch.cuda.stream(side):
                if not ticked[0]:
                    opt.tick()                                     # the optimiser's step count advances off the critical path
                    ticked[0] = True
                model.plan_sampling(xn, (s1n, s2n), out=plan_out)     # the kernels write the other graph's plan buffers in place

        def mark():                                                # sa2late / sa2end: only the dependency point is taken here
            fork_ev[0] = torch.cuda.Event()
            fork_ev[0].record(main)

        # (N > 1: the two stages are two graphs and a fork must be joined inside the graph that opened it, so the sampling
        # branch belongs to stage 2 -- the SA2 + SA1 backward, 1.4 ms -- there)
        if plan_out is not None and args.fork == "start" and not use_dist:
            fork()
        gate_at = None
        if side_graph and graph_state.get("capturing_main"):
            def open_gate():                       # from here on the other stream's pyramid may start (--fork: behind SA2 by default); the same one-thread
                gate_open(opt.t_dev if not ticked[0] else None)     # launch advances the optimiser's device step count (the update is the step's last node)
                ticked[0] = True
            gate_at = args.fork
            if gate_at == "start":
                open_gate()
        tap = {} if use_dist else None
        if args.diag_fixed_plan:
            plan_in, plan_out = graph_state["fixed_plan"], None
        loss, _ = model(x, (s1, s2), plan=plan_in, tap=tap,
                        after_sa2=cut if cut is not None else (open_gate if gate_at == "sa2" else ((fork if args.fork == "sa2" else (mark if args.fork in ("sa2late", "sa2end") else None)) if plan_out is not None and not use_dist else None)),
                        after_sa1=open_gate if gate_at == "sa1" else None,
                        after_sa3=open_gate if gate_at == "sa3" else (((fork if args.fork == "sa3" else ((lambda: fork(fork_ev[0])) if args.fork == "sa2late" else None)) if plan_out is not None and not use_dist else None)), labels=y)
        if not use_dist:
            if gate_at == "loss":
                open_gate()
            if plan_out is not None and args.fork == "loss":
                fork()
            loss.backward(ONE)
            if plan_out is not None and args.fork == "sa2end":
                fork(fork_ev[0])
            if plan_out is not None:
                main.wait_stream(side)             # join: the branch is part of this step
            if ADAM_IN_GRAPH:
                if not ticked[0]:
                    opt.tick()
                finish(None)                       # the update is the last node of the captured step
            return loss, None, None, None
        l2 = tap["l2_points"]
        (g_l2,) = torch.autograd.grad(loss, [l2], [ONE])  # the kernels write the head's and SA3's gradients straight into the flat views
        return loss, l2, g_l2, (fork if plan_out is not None else None)

    def stage2(l2, g_l2, fork):
        if fork is not None:
            fork()
        torch.autograd.backward([l2], [g_l2])      # SA2 + SA1
        if fork is not None:
            main.wait_stream(side)


    def capture():
        """Returns True when the graph(s) were captured; on any capture failure the bench falls back to eager launches."""
        if ZERO_IN_ADAM:
            flat.zero_grad()                       # (the passes ahead of a capture may have ended without an optimiser step)
        torch.cuda.synchronize()
        try:
            load_slot(0, cur["j"])                 # the first replayed step (set 0) trains on batch cur["j"]: its inputs and, below, its plan
            x0, _, s10, s20 = unpack(slots[0])
            if args.diag_fixed_plan:
                graph_state["fixed_plan"] = model.plan_sampling(x0, (s10, s20))
                torch.cuda.synchronize()
            n_sets = 2 if args.overlap else 1
            graph_state["prev_end"] = None
            bufs = None
            if args.overlap:
                p0 = model.plan_sampling(x0, (s10, s20))
                bufs = [tuple(tuple(t.clone() for t in lvl) for lvl in p0) for _ in range(2)]
                torch.cuda.synchronize()
            gs, losses = [], []
            if split_pyr:
                bufs.append(tuple(tuple(t.clone() for t in lvl) for lvl in p0))      # a third set
                gB, gA = [], []
                for i in range(3):
                    g1 = torch.cuda.CUDAGraph()
                    graph_state["capturing_main"] = True
                    with torch.cuda.graph(g1, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                        loss, _, _, _ = stage1(bufs[i], None, si=i)
                    graph_state["capturing_main"] = False
                    gs.append((g1, None, None))
                    losses.append(loss)
                    for strm, lst, stg, k, slot_w in ((side, gB, "rest", (i + 1) % 3, 0), (side2, gA, "fps1", (i + 2) % 3, 1)):
                        g2 = torch.cuda.CUDAGraph()
                        strm.wait_stream(main)
                        with torch.cuda.stream(strm):
                            with torch.cuda.graph(g2, stream=strm, capture_error_mode="thread_local"):
                                gate_wait(slot_w)
                                if stg == "fps1":
                                    stage_to_slot(stage_in2, k)
                                xk, _, s1k, s2k = unpack(slots[k])
                                model.plan_sampling(xk, (s1k, s2k), out=bufs[k], stage=stg)
                        main.wait_stream(strm)
                        lst.append(g2)
                # the pipeline's first step needs batch j + 1 in its slot and its first-level centroids in its buffers (what graph A of a
                # previous step would have left there)
                load_slot(1, cur["j"] + 1)
                x1, _, s11, s21 = unpack(slots[1])
                model.plan_sampling(x1, (s11, s21), out=bufs[1], stage="fps1")
                graph_state["g"], graph_state["loss"], graph_state["bufs"], graph_state["gB"], graph_state["gA"] = gs, losses, bufs, gB, gA
                graph_state["gside"] = None
                graph_state["evB"], graph_state["evA"] = [None] * 3, [None] * 3
                torch.cuda.synchronize()
                return True
            if side_graph:
                gside = []
                for i in range(2):
                    g1 = torch.cuda.CUDAGraph()
                    graph_state["capturing_main"] = True
                    with torch.cuda.graph(g1, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                        loss, _, _, _ = stage1(bufs[i], None, si=i)
                    graph_state["capturing_main"] = False
                    gs.append((g1, None, None))
                    losses.append(loss)
                    g2 = torch.cuda.CUDAGraph()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        with torch.cuda.graph(g2, stream=side, capture_error_mode="thread_local"):
                            gate_wait()
                            stage_to_slot(stage_in, 1 - i)
                            xn, _, s1n, s2n = unpack(slots[1 - i])
                            if not args.diag_empty_side:
                                model.plan_sampling(xn, (s1n, s2n), out=bufs[1 - i])
                    main.wait_stream(side)
                    gside.append(g2)
                graph_state["g"], graph_state["loss"], graph_state["bufs"], graph_state["gside"] = gs, losses, bufs, gside
                graph_state["side_ev"] = [None, None]
                torch.cuda.synchronize()
                return True
            for i in range(n_sets):
                pin, pout = (bufs[i], bufs[1 - i]) if args.overlap else (None, None)
                if ext_sampling:
                    pout = None                    # filled from outside the graph (step())
                g1 = torch.cuda.CUDAGraph()
                g1b = g2 = None
                if dist_side_graph:
                    # two graphs per step (one memory pool): forward + loss + the head's and SA3's backward | SA2 + SA1 backward, the tail bucket's
                    # all-reduce issued between them; the pyramid as a third graph on the side stream, gated behind SA2 on the device
                    g2 = torch.cuda.CUDAGraph()
                    g1.capture_begin(capture_error_mode="thread_local")
                    try:
                        loss, l2, g_l2, _ = stage1(pin, None, (lambda: gate_open(None)), si=i)
                        g1.capture_end()
                        g2.capture_begin(pool=g1.pool(), capture_error_mode="thread_local")
                        stage2(l2, g_l2, None)
                        g2.capture_end()
                    except Exception:
                        for g in (g1, g2):
                            try:
                                g.capture_end()
                            except Exception:     # noqa: BLE001
                                pass
                        raise
                    gsd = torch.cuda.CUDAGraph()
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        with torch.cuda.graph(gsd, stream=side, capture_error_mode="thread_local"):
                            gate_wait()
                            stage_to_slot(stage_in, 1 - i)
                            xn, _, s1n, s2n = unpack(slots[1 - i])
                            model.plan_sampling(xn, (s1n, s2n), out=bufs[1 - i])
                    main.wait_stream(side)
                    graph_state.setdefault("gside_dist", [None, None])[i] = gsd
                    graph_state["side_ev"] = [None, None]
                    gs.append((g1, None, g2))
                    losses.append(loss)
                    continue
                if ext_sampling:
                    # three graphs per step (one memory pool): forward up to SA2 | SA3 + head + their backward | SA2 + SA1 backward.
                    # The first cut is where the side stream's sampling is released (the SA3 / head kernels are small grids that
                    # leave CUs free, as with the in-graph fork), the second is where the tail bucket's all-reduce is issued.
                    g1b, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()

                    def cut():
                        g1.capture_end()
                        g1b.capture_begin(pool=g1.pool(), capture_error_mode="thread_local")

                    g1.capture_begin(capture_error_mode="thread_local")
                    try:
                        loss, l2, g_l2, fork = stage1(pin, None, cut, si=i)
                        g1b.capture_end()
                        g2.capture_begin(pool=g1.pool(), capture_error_mode="thread_local")
                        stage2(l2, g_l2, None)
                        g2.capture_end()
                    except Exception:
                        for g in (g1, g1b, g2):           # leave no capture open behind a failure
                            try:
                                g.capture_end()
                            except Exception:     # noqa: BLE001
                                pass
                        raise
                    gs.append((g1, g1b, g2))
                    losses.append(loss)
                    continue
                # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
                with torch.cuda.graph(g1, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                    loss, l2, g_l2, fork = stage1(pin, pout, si=i)
                if use_dist:   # stage 2 is a second graph (same memory pool): the collective of the tail bucket goes between them
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, pool=g1.pool(), stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                        stage2(l2, g_l2, fork)
                gs.append((g1, None, g2))
                losses.append(loss)
            graph_state["g"], graph_state["loss"], graph_state["bufs"] = gs, losses, bufs
        except Exception as e:   # noqa: BLE001
            graph_state["why"] = "%s: %s" % (type(e).__name__, e)
            if args.require_graph:
                raise SystemExit("[bench] hipGraph capture failed (%s) and --require-graph is set" % graph_state["why"])
            print("[bench] WARNING: hipGraph capture failed (%s); the timed steps launch EAGERLY -- the slow path" % graph_state["why"], file=sys.stderr)
            graph_state["g"], graph_state["loss"] = None, None
            torch.cuda.synchronize()
            return False
        return True

    def finish(work):
        """reduce what is left of the bucket, then the optimiser"""
        if use_dist:
            if work is not None:
                work.wait()                        # the main stream waits for the tail bucket's collective (no host block)
            scale = flat.allreduce_grads(0, split)
        else:
            scale = flat.allreduce_grads()
        if args.dump_trajectory:
            if state["last_grad"] is None:
                state["last_grad"] = torch.empty_like(flat.grad)
            state["last_grad"].copy_(flat.grad)                 # (tests: the optimiser clears the bucket)
        if ADAM_IN_GRAPH:
            opt.step_dev(scale, zero_grad=ZERO_IN_ADAM)
        else:
            opt.step(scale, zero_grad=ZERO_IN_ADAM)

    def sample_into(plan_out, si):
        """the next batch's pyramid (slot si), enqueued on the side stream beside the graph that is being replayed (N > 1)"""
        xn, _, s1n, s2n = unpack(slots[si])
        with torch.cuda.stream(side):
            model.plan_sampling(xn, (s1n, s2n), out=plan_out)

    PREV_END_MODE = os.environ.get("PAPC_PREV_END_MODE", "timing")

    class _HipEvent:
        """a raw HIP event with flags torch does not expose (A/B of what the cross-stream wait costs)"""
        _hip = None

        def __init__(self):
            import ctypes as C
            if _HipEvent._hip is None:
                # the HIP runtime torch itself loaded (its own copy: a second instance would know nothing of torch's streams)
                path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l), "libamdhip64.so")
                _HipEvent._hip = C.CDLL(path)
            self.h = C.c_void_p()
            rc = _HipEvent._hip.hipEventCreateWithFlags(C.byref(self.h), C.c_uint(0x2 | 0x40000000))     # DisableTiming | ReleaseToDevice
            assert rc == 0, rc

        def record(self, stream):
            import ctypes as C
            rc = _HipEvent._hip.hipEventRecord(self.h, C.c_void_p(stream.cuda_stream))
            assert rc == 0, rc
            return self

        def wait_on(self, stream):
            import ctypes as C
            rc = _HipEvent._hip.hipStreamWaitEvent(C.c_void_p(stream.cuda_stream), self.h, C.c_uint(0))
            assert rc == 0, rc

        def __del__(self):
            try:
                _HipEvent._hip.hipEventDestroy(self.h)
            except Exception:     # noqa: BLE001
                pass

    def wait_prev_end(stream):
        pe = graph_state.get("prev_end")
        if pe is None:
            return
        if isinstance(pe, _HipEvent):
            pe.wait_on(stream)
        else:
            stream.wait_event(pe)

    # (diagnostics with --diag-empty-side only: pieces of the two-stream structure switched off one at a time -- results are then garbage)
    DIAG_NO_SLOT_COPY = os.environ.get("PAPC_DIAG_NO_SLOT_COPY") == "1"
    DIAG_NO_SIDE_EV = os.environ.get("PAPC_DIAG_NO_SIDE_EV") == "1"
    DIAG_NO_PREV_END = os.environ.get("PAPC_DIAG_NO_PREV_END") == "1"

    def side_replay(g, i, j):
        """The gated pyramid graph of set i on the side stream: batch j + 1 goes into the staging pair (eagerly, in stream order behind the previous
        replay), the graph copies it on into slot 1 - i behind its gate and fills bufs[1 - i] from it.  Slot and plan buffers were last read by the
        PREVIOUS step's kernels (set 1 - i: its backward reads the grouping lists to the end); what orders the graph behind them is the device-side gate,
        which this step opens behind its SA2 -- in stream order behind the whole previous step -- and which COUNTS its openings: a wait that gives
        up (after ~2 s) is a sticky error that makes the run abort (sync()), never a silently mis-ordered step (round 5: a give-up after 35 ms "started
        anyway" and shifted every later pyramid one opening early).  --side-events adds the belt to the braces: the side stream also waits for the
        previous step's end-of-step event before anything is enqueued -- measured at 25-31 us per step, hence opt-in."""
        with torch.cuda.stream(side):
            if args.side_events and not DIAG_NO_PREV_END:
                wait_prev_end(side)
            if not DIAG_NO_SLOT_COPY:
                load_stage(stage_in, j + 1)           # (the graph copies it on into slot 1 - i behind its gate)
            g.replay()                             # gated on the device: starts when this step has enqueued SA2
            ev = torch.cuda.Event()
            ev.record(side)
        graph_state["side_ev"][1 - i] = None if DIAG_NO_SIDE_EV else ev

    def step_graph():
        i = graph_state["i"] % len(graph_state["g"])
        graph_state["i"] += 1
        j = cur["j"]
        cur["j"] = j + 1
        g1, g1b, g2 = graph_state["g"][i]
        if split_pyr and graph_state.get("gB"):
            ev = graph_state["evB"][i]
            if ev is not None:
                main.wait_event(ev)                # slot i and bufs[i] are complete: graph B of the previous step (which waited for graph A of the one before)
            g1.replay()
            prev_end = graph_state.get("prev_end")
            with torch.cuda.stream(side):          # B: the rest of batch j + 1's pyramid, from the centroids graph A left in bufs[i + 1] one step ago
                if prev_end is not None and args.side_events:
                    side.wait_event(prev_end)
                if graph_state["evA"][(i + 1) % 3] is not None:
                    side.wait_event(graph_state["evA"][(i + 1) % 3])
                graph_state["gB"][i].replay()
                ev = torch.cuda.Event()
                ev.record(side)
                graph_state["evB"][(i + 1) % 3] = ev
            with torch.cuda.stream(side2):         # A: batch j + 2 into its slot, its first-level farthest-point sampling into bufs[i + 2]
                if prev_end is not None and args.side_events:
                    side2.wait_event(prev_end)     # (their last reader: the previous step, set i + 2 = i - 1)
                load_stage(stage_in2, j + 2)          # (graph A copies it on into slot i + 2 behind its gate)
                graph_state["gA"][i].replay()
                ev = torch.cuda.Event()
                ev.record(side2)
                graph_state["evA"][(i + 2) % 3] = ev
            return graph_state["loss"][i]
        if side_graph and graph_state.get("gside"):
            ev = graph_state["side_ev"][i]
            if ev is not None:
                main.wait_event(ev)                # slot i and bufs[i] were filled by the side stream during the previous step
            g1.replay()
            side_replay(graph_state["gside"][i], i, j)
            return graph_state["loss"][i]
        if dist_side_graph and graph_state.get("gside_dist"):
            ev = graph_state["side_ev"][i]
            if ev is not None:
                main.wait_event(ev)
            g1.replay()
            side_replay(graph_state["gside_dist"][i], i, j)
            _, work = flat.allreduce_grads(split, None, async_op=True)
            g2.replay()
            finish(work)
            return graph_state["loss"][i]
        # the other structures load the inputs on the main stream ahead of the replay: the next batch (a forked branch of this graph, or the eager
        # side-stream sampling behind it, reads slot 1 - i), or -- in-line sampling, one graph -- this step's own
        if args.overlap:
            load_slot(1 - i, j + 1)
        else:
            load_slot(0, j)
        if ext_sampling:
            main.wait_stream(side)                 # this batch's plan (bufs[i]) was filled on the side stream during the last step
        g1.replay()
        if ext_sampling:
            side.wait_stream(main)                 # released when the main stream reaches SA3; bufs[1 - i]'s last reader is long done
            sample_into(graph_state["bufs"][1 - i], 1 - i)
            g1b.replay()
        work = None
        if g2 is not None:
            _, work = flat.allreduce_grads(split, None, async_op=True)
            g2.replay()
        if not ADAM_IN_GRAPH:
            finish(work)
        return graph_state["loss"][i]

    step_events = []                               # (timed region) one end-of-step event per step

    def step():
        if stall_gate is not None and graph_state.get("timing") and len(step_events) % 3 == 2:
            # (tests) nobody opens this gate: the launch spins for ~diag_stall_ms on the main stream, then gives up
            _lib.check(lib.papc_flag_wait(stall_gate.data_ptr(), int(args.diag_stall_ms * 1150), main.cuda_stream), "papc_flag_wait")
        loss = step_eager() if graph_state["g"] is None else step_graph()
        end = torch.cuda.Event(enable_timing=True)
        end.record(main)
        if PREV_END_MODE == "plain":               # a second, non-timing event for the cross-stream wait
            pe = torch.cuda.Event()
            pe.record(main)
            graph_state["prev_end"] = pe
        elif PREV_END_MODE == "hipdev":            # ... created with hipEventDisableTiming | hipEventReleaseToDevice (device-scope release)
            graph_state["prev_end"] = _HipEvent().record(main)
        else:
            graph_state["prev_end"] = end          # what the next step's side graph waits for before it touches the buffers this step read
        if graph_state.get("timing"):
            step_events.append(end)
        return loss


    gate_timeouts = [0]

    def sync():
        torch.cuda.synchronize()
        if gate is not None:
            n = int(gate[2].item())
            if n != gate_timeouts[0]:
                gate_timeouts[0] = n
                msg = "[bench] %d device-side gate wait(s) gave up after PAPC_GATE_SPINS=%d sleeps: the sampling graph started without its opening" % (n, GATE_SPINS)
                if not args.allow_gate_timeout:
                    raise SystemExit(msg + " -- aborting (the plan buffers stay ordered by stream events, but the measured overlap is not the designed one)")
                print(msg, file=sys.stderr)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup, then an untimed 3-step pass with the event profiler on every family to find the dominant one
    # N = 1: the W warm-up steps, then the family pass, run eagerly ahead of the capture.  N > 1: the passes ahead of the capture are
    # forward + backward only (no collective on the stream about to be captured, see above); the W warm-up steps proper -- gradient
    # exchange + Adam -- follow the capture, on the launch structure that is timed.
    loss = None
    pre = (lambda: step_eager(exchange=False)) if use_dist else step
    for _ in range(max(1, args.warmup) if not use_dist else 2):
        loss = pre()
    torch.cuda.synchronize()
    lib.papc_prof_enable(0x3FF)
    lib.papc_prof_reset()
    NPROF = 3
    for _ in range(NPROF):
        loss = pre()
    torch.cuda.synchronize()
    fam = prof_read(lib)
    lib.papc_prof_enable(0)
    dominant = max((k for k in fam if k != 9), key=lambda k: fam[k][0])
    if args.profile_all and rank == 0:
        tot = sum(v[0] for v in fam.values())
        for k, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
            print("  %-14s %8.3f ms/step  %4d launches/step  %5.1f%%" % (K_NAMES[k], ms / NPROF, n // NPROF,
                                                                         100.0 * ms / max(tot, 1e-9)), file=sys.stderr)
    if use_graph:
        loss = None                               # drop the last eager autograd graph before capturing
        use_graph = capture()                     # event profiler off: nothing but kernels, memsets and copies in the graph
    for _ in range(2 + (max(1, args.warmup) if use_dist else 0)):   # (also without a graph: every launch structure runs the same number of optimiser steps)
        loss = step()
    torch.cuda.synchronize()
    if args.dry_run:                              # launch-structure check only (no timing): what got captured, on how many ranks
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "graphs_per_step": (len([g for g in graph_state["g"][0] if g is not None]) if use_graph else 0),
                              "graph_sets": len(graph_state["g"]) if use_graph else 0, "capture_error": graph_state["why"],
                              "sampling": ("gated hipGraph on the side stream" if dist_side_graph else "side stream beside the graph replays") if ext_sampling else ("in-graph fork" if (use_graph and args.overlap) else "in-line"),
                              "loss": float(loss.item())}))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not use_graph:
        lib.papc_prof_enable(1 << dominant)       # eager timed region: event pairs only around the dominant family
        lib.papc_prof_reset()

    # ---- timed region
    traj = []

    def timed_region(step_ms=None):
        """EXACTLY --steps steps between barrier + synchronize on both sides (max over ranks); ``step_ms`` (a list) receives every step's duration
        from device events: end-of-step event to end-of-step event on the main stream (the first from an event recorded behind the opening sync)"""
        sync()
        del step_events[:]
        j0 = cur["j"]
        start = torch.cuda.Event(enable_timing=True)
        start.record(main)
        graph_state["timing"] = True
        t0 = time.perf_counter()
        ls = None
        for _ in range(args.steps):
            ls = step()
            if args.dump_trajectory:
                traj.append(ls.detach().clone())  # (the replayed graphs overwrite their loss tensor)
        sync()
        el = time.perf_counter() - t0
        graph_state["timing"] = False
        if step_ms is not None:
            evs = [start] + step_events
            step_ms.extend(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
            step_batch.extend((j0 + k) % NB for k in range(len(step_events)))
        if use_dist:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ls

    def stats_of(ms):
        ms = sorted(ms)
        n = len(ms)
        return {"ms_median": round(0.5 * (ms[(n - 1) // 2] + ms[n // 2]), 4), "ms_min": round(ms[0], 4), "ms_p95": round(ms[min(n - 1, int(0.95 * n))], 4),
                "steps_timed": n}

    def more_windows(step_ms):
        """further windows of --steps steps until the per-step statistics stand on --min-timed-steps steps (the contract window stays the first)"""
        while len(step_ms) < args.min_timed_steps and not args.dump_trajectory and not args.diag_stall_ms:
            timed_region(step_ms)

    step_ms, step_batch = [], []               # per-step durations (device events) and the batch each step trained on
    elapsed, loss = timed_region(step_ms)
    final_loss = float(loss.item())
    more_windows(step_ms)
    if args.dump_trajectory:                      # rank 0 -> FILE, rank r -> FILE.rank<r>.npz (tests: the replicas must stay identical)
        import numpy as np
        np.savez(args.dump_trajectory if rank == 0 else "%s.rank%d.npz" % (args.dump_trajectory, rank), loss=torch.stack(traj).cpu().numpy(), params=flat.data.detach().cpu().numpy(), params0=params0, grad=(state["last_grad"] if state["last_grad"] is not None else flat.grad).detach().cpu().numpy(),
                 graph=np.array(int(use_graph)), overlap=np.array(int(args.overlap)))
    n_roof = args.steps
    if use_graph:
        # kernels inside a replayed graph cannot carry host-visible event pairs: the dominant family's launch durations are
        # measured on the same kernels, same inputs, launched eagerly for min(K, 20) further steps right after the timed region
        n_roof = min(args.steps, 20)
        lib.papc_prof_enable(1 << dominant)
        lib.papc_prof_reset()
        for _ in range(n_roof):
            step_eager()
        torch.cuda.synchronize()
    # what an event pair itself costs: the two markers are barrier packets with a timestamp write each, and a pair around NOTHING already reads
    # several microseconds -- measured here on the same stream (median of 200 empty pairs) and reported beside the raw figure; `roofline.frac` is
    # computed on the family's time with that overhead taken off every launch, which is what agrees with rocprofv3's per-kernel durations
    # (profiles/), `frac_raw_pairs` keeps the uncorrected one.  (Keeping the host a whole step ahead of the device behind a spinning kernel, so
    # that no pair sees launch latency, was tried first: the family's time did not move -- the overhead is the markers', not the host's.)
    pair_ms = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        e1.record(main)
        pair_ms.append((e0, e1))
    torch.cuda.synchronize()
    pair_ms = sorted(a.elapsed_time(b) for a, b in pair_ms)
    pair_overhead_ms = pair_ms[len(pair_ms) // 2]
    dom_ms, dom_n = prof_read(lib)[dominant]
    lib.papc_prof_enable(0)
    assert final_loss == final_loss, "loss is NaN"

    # ---- how much of the headline depends on the generator: the compacted SA2 stack (distinct neighbours only, papc_amd/compact.py) pays
    # where the ball query's padding copies are plentiful (pointnet2_basic_layers.py:118-124) -- 0.56 of SA2's rows are kept on the SURVEY 8d
    # clouds, 0.91-0.99 on ShapeNet-like surfaces, where the auto policy stays padded.  Same process, same box, same weights trajectory
    # continued: the timed region once more with the stack forced padded (`value_padded`), and the measured row fraction.
    from papc_amd import stack as _stack
    plans_used = dict(_stack.LAST_PLANS)
    rows_sa2, row_fraction = None, 1.0
    sa2_key = (B * 128 * 64, (128, 128, 256))
    compact_on = bool(plans_used.get(sa2_key, {}).get("compact"))
    row_fractions = []
    if compact_on:
        for k in range(NB):                       # the compacted stack's device-side row count, batch by batch (the step cycles through them)
            xb, _, s1b, s2b = bt(k)
            pl2 = model.plan_sampling(xb, (s1b, s2b))[1]
            if len(pl2) in (9, 12):
                row_fractions.append(int(pl2[4][0].item()) / float(B * 128 * 64))
        if row_fractions:
            row_fraction = sum(row_fractions) / len(row_fractions)
            rows_sa2 = int(round(row_fraction * B * 128 * 64))
    padded = None
    if compact_on and not use_dist and not args.diag_fixed_plan and not args.dump_trajectory and not args.no_padded_leg:     # (one process: a second capture behind collectives would trip RCCL's watchdog, see `comm` above)
        model.sa2.compact = False                 # forced padded (layers.PointNetSetAbstraction._compact_mode)
        state["plan"], state["ev"] = None, None
        graph_state.update({"g": None, "loss": None, "i": 0})
        for _ in range(3):
            loss_p = step()                       # eager: allocations of the padded shapes
        torch.cuda.synchronize()
        loss_p = None
        if not args.no_graph:
            capture()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ms_p = []
        n_main = len(step_batch)
        el_p, loss_p = timed_region(ms_p)
        more_windows(ms_p)
        assert not _stack.LAST_PLANS.get(sa2_key, {}).get("compact"), "the padded leg still ran the compacted stack"
        del step_batch[n_main:]                    # (the padded leg's steps are not part of the per-batch medians below)
        padded = {"value": round(B * args.steps / el_p, 2), "ms_per_step": round(1e3 * el_p / args.steps, 3),
                  "graph": graph_state["g"] is not None, "stats": stats_of(ms_p)}
        model.sa2.compact = None

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        flop_fixed, byts_fixed = algorithmic_work(B, N).get(dominant, (0.0, 0.0))
        # priced on what the launches process: the path each stack took (papc_sa_mlp_plan) and the compacted stack's device-side row count;
        # the SURVEY 8d figure, fixed across rounds, stays beside it as `frac_fixed_bytes`
        flop, byts = moved_work(B, N, plans_used, rows_sa2).get(dominant, (flop_fixed, byts_fixed))
        per_step_raw_s = (dom_ms / 1e3) / n_roof if dom_ms > 0 else float("nan")
        per_step_s = per_step_raw_s - (dom_n / max(1, n_roof)) * pair_overhead_ms / 1e3       # event-pair overhead off every launch (see above)
        # which roof bounds the family: its matrix time at the rate the instruction mix allows (exact 3-way bf16 split =
        # 6 bf16 MFMA products per fp32 product -> 2500 / 6 TFLOP/s of algorithmic fp32 work) against its HBM time at 8 TB/s
        f32_exact = os.environ.get("PAPC_GEMM_F32") == "1" and os.environ.get("PAPC_DW_F32") == "1"
        mfma_peak = PEAK_MFMA_F32_TFLOPS if f32_exact else PEAK_MFMA_BF16_TFLOPS / 6.0
        t_mfma = flop / (mfma_peak * 1e12)
        t_hbm = byts / (PEAK_HBM_GBS * 1e9)
        if flop > 0 and t_mfma >= t_hbm:
            achieved = flop / per_step_s / 1e12
            roof = {"bound": "mfma", "kernel": K_NAMES[dominant], "achieved": round(achieved, 2), "peak": round(mfma_peak, 1),
                    "unit": "TFLOP/s", "frac": round(achieved / mfma_peak, 4), "traffic": None}
        else:
            achieved = byts / per_step_s / 1e9
            roof = {"bound": "hbm", "kernel": K_NAMES[dominant], "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
        # HBM traffic of the family from the PMC counters: they need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE
        # separately, --kernel-trace only), so they are collected offline (tools/refresh_profiles.sh) and read back from the
        # committed per-family summary of the latest round
        roof["traffic_note"] = "no profiles/*_bench_pmc_family.json found"
        import glob
        fams = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_bench_pmc_family.json")))
        if fams:
            try:
                with open(fams[-1]) as fh:
                    rec = json.load(fh).get(K_NAMES[dominant])
                if rec:
                    lps = max(1, dom_n // n_roof)
                    if "traffic_MB_per_step" in rec:     # per step over THIS run's launches per step (an entry point may launch several kernels)
                        rec = dict(rec, traffic_MB_per_launch=rec["traffic_MB_per_step"] / lps)
                    roof["traffic"] = round(rec["traffic_MB_per_launch"] * 1e6)     # bytes per launch, like `achieved`
                    # the same fraction on COUNTER bytes (what the family really moved through HBM in the profiled run) beside the one on the
                    # fixed SURVEY 8d algorithmic bytes: the latter still counts tensors that later rounds stopped storing
                    if roof["bound"] == "hbm" and per_step_s == per_step_s:
                        roof["frac_counter_bytes"] = round(rec["traffic_MB_per_launch"] * 1e6 * lps / per_step_s / (PEAK_HBM_GBS * 1e9), 4)
                    roof["traffic_note"] = ("%s: 2 x FETCH_SIZE + WRITE_SIZE per launch of the family (separate rocprofv3 --pmc passes, gfx950 "
                                            "FETCH_SIZE correction x2), %.1f MB against %.1f MB algorithmic per launch"
                                            % (os.path.join("profiles", os.path.basename(fams[-1])), rec["traffic_MB_per_launch"],
                                               byts / 1e6 / max(1, dom_n // n_roof)))
            except (OSError, ValueError, KeyError) as e:
                roof["traffic_note"] = "could not read %s: %s" % (fams[-1], e)
        roof["algorithmic"] = {"GFLOP_per_step": round(flop / 1e9, 2), "MB_per_step": round(byts / 1e6, 1),
                               "mfma_floor_ms": round(t_mfma * 1e3, 3), "hbm_floor_ms": round(t_hbm * 1e3, 3),
                               "basis": "operands the family's launches process on the paths taken (bench.py::moved_work): moment first layer, "
                                        "no-store max layer, gather-add first layer, compacted rows = %s" % (rows_sa2 if rows_sa2 else "padded"),
                               "fixed_MB_per_step": round(byts_fixed / 1e6, 1), "fixed_GFLOP_per_step": round(flop_fixed / 1e9, 2)}
        if roof["bound"] == "hbm" and per_step_s == per_step_s:
            roof["frac_fixed_bytes"] = round(byts_fixed / per_step_s / (PEAK_HBM_GBS * 1e9), 4)
        roof["launches_per_step"] = dom_n // n_roof
        roof["avg_launch_ms"] = round(1e3 * per_step_s / max(1, dom_n // n_roof), 4)
        roof["ms_per_step"] = round(1e3 * per_step_s, 3)
        roof["ms_per_step_raw_pairs"] = round(dom_ms / n_roof, 3)
        roof["event_pair_overhead_us"] = round(1e3 * pair_overhead_ms, 2)
        if roof["bound"] == "hbm" and per_step_raw_s == per_step_raw_s:
            roof["frac_raw_pairs"] = round(byts / per_step_raw_s / (PEAK_HBM_GBS * 1e9), 4)
        roof["timing"] = ("HIP event pairs around every launch of the family, %d eager steps right after the graph-replayed "
                          "timed region; the overhead of an event pair itself (median of 200 empty pairs on the same stream) is taken off every "
                          "launch, frac_raw_pairs / ms_per_step_raw_pairs keep the uncorrected figures" % n_roof) if use_graph else "HIP event pairs around every launch of the family over the timed region"

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(N)
        out = {
            "metric": "point-clouds/sec (fwd+bwd) PointNet++SSG B=32 N=4096" + (" -- DIAGNOSTIC, sampling excluded: not a benchmark value" if (args.diag_fixed_plan or args.diag_empty_side) else ""),
            "value": round(value, 2), "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), **stats_of(step_ms), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PointNet++SSG classify fwd+bwd+Adam, B=%d clouds/GPU, N=%d (BASELINE configs[1])" % (B, N),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "final_loss": round(final_loss, 4),
                       "compact_row_fraction": round(row_fraction, 4), "compact_row_fraction_per_batch": [round(f, 4) for f in row_fractions],
                       "ms_median_per_batch": [round(stats_of([m for m, bi in zip(step_ms, step_batch) if bi == k])["ms_median"], 4) if any(bi == k for bi in step_batch) else None
                                               for k in range(NB)],
                       "clouds_per_s_per_batch": [round(world * B / (1e-3 * stats_of([m for m, bi in zip(step_ms, step_batch) if bi == k])["ms_median"]), 1)
                                                  if any(bi == k for bi in step_batch) else None for k in range(NB)],      # (batch 0 = the one batch rounds 1-5 timed)
                       "batches": "%d distinct resident batches of B clouds (seeds 1234 + rank + 100003 k), step j trains on batch j mod %d; the next batch is loaded into the "
                                  "graphs' input slot and sampled on the side stream" % (NB, NB),
                       "timing": "value / ms_per_step: wall clock around the first window of --steps steps between barrier + synchronize (max over ranks); ms_median / ms_min / "
                                 "ms_p95: device events around every step (end-of-step event to end-of-step event on the launch stream, rank 0) over steps_timed steps",
                       "gate_timeouts": gate_timeouts[0],
                       "compact": ("SA2 on its distinct neighbours (auto policy: kept rows %.3f of the padded count on this generator; ShapeNet-like "
                                   "surfaces keep 0.91-0.99 and stay padded -> `value_padded` is their rate)" % row_fraction) if compact_on else "padded (policy)",
                       "sampling": ("software-pipelined: batch i+1's FPS + ball-query pyramid runs as a second branch (side stream) of "
                                    "batch i's step, %s; every timed step computes one full pyramid"
                                    % (("a hipGraph of its own on the side stream, gated on the device behind SA2 (papc_flag_set / papc_flag_wait)" if dist_side_graph else "enqueued on the side stream beside the graph replay") if ext_sampling else
                                       (("two hipGraphs on two side streams, no graph edge to the step's: the first level's farthest-point sampling of the batch TWO steps "
                                         "ahead beside the rest of the NEXT batch's pyramid (three alternating sets), both " if split_pyr else
                                         "a second hipGraph on the side stream, no graph edge to the step's (a forked branch costs the main chain ~60 us per replay): ") +
                                        "gated on the device behind %s (papc_flag_set / papc_flag_wait), plan buffers ordered by stream events" % args.fork if (side_graph and use_graph)
                                        else "fork at " + args.fork))) if args.overlap else "in-line",
                       "mfma": "fp32 operands as exact 3-way bf16 splits, 6 v_mfma_f32_32x32x16_bf16 per 32x32x16 block, fp32 "
                               "accumulate (PAPC_GEMM_F32=1 PAPC_DW_F32=1 select v_mfma_f32_32x32x2_f32); gather-layer dW stays on the f32 MFMA",
                       "launch": ("hipGraph replay of fwd+loss+bwd (%d graph(s) per step, two alternating sets), %s"
                                  % (len([g for g in graph_state["g"][0] if g is not None]),
                                     "Adam (which also clears the gradient bucket) as the graph's last node" if ADAM_IN_GRAPH
                                     else "eager all-reduce + Adam (which also clears the gradient bucket)")) if use_graph
                                 else ("eager" + (" (hipGraph capture FAILED: %s)" % graph_state["why"] if graph_state["why"] else "")),
                       "collectives": ("world %d, backend %s (RCCL %s); two-stage backward: all-reduce of the [SA3 | FC head] tail of the flat bucket "
                                       "(%d floats) in flight during the SA2 + SA1 backward, then the %d-float head of the bucket; RCCL over xGMI, "
                                       "per-GPU BatchNorm statistics"
                                       % (world, dist.get_backend(), ".".join(str(v) for v in torch.cuda.nccl.version()), flat.numel - split, split))
                                      if use_dist else "none (one process)"},
            "value_padded": padded["value"] if padded else None,
            "ms_per_step_padded": padded["ms_per_step"] if padded else None,
            "padded_stats": padded["stats"] if padded else None,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(N):
    """torch-CPU port of the reference's op decomposition, one full train step, on this box's host cores.
    Bounded: a 2-cloud probe decides whether the full B=32 step fits the ~10-30 s budget."""
    from oracle import torch_cpu_reference as T
    cores = len(os.sched_getaffinity(0))
    threads = min(cores, 64)
    t_probe, _ = T.time_train_step(2, N, threads)
    Bs = 32 if t_probe * 16 < 45 else (8 if t_probe * 4 < 45 else 2)
    if Bs == 2:
        t, cps = t_probe, 2 / t_probe
        n_steps = 1
    else:
        times = [T.time_train_step(Bs, N, threads)[0]]
        while len(times) < 3 and sum(times) + times[-1] < 20.0:      # up to three steps inside the ~10-30 s budget: median
            times.append(T.time_train_step(Bs, N, threads)[0])
        t = sorted(times)[len(times) // 2]
        cps, n_steps = Bs / t, len(times)
    return {"value": round(cps, 3), "unit": "point-clouds/s", "cores": threads, "kind": "port",
            "sample": "median of %d fwd+bwd+Adam step(s) of the torch-CPU transliteration (oracle/torch_cpu_reference.py), B=%d N=%d, "
                      "%.1f s per step" % (n_steps, Bs, N, t)}


if __name__ == "__main__":
    main()
