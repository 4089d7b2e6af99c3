#!/bin/bash
# Collect the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# then summarise locally with tools/refresh_profiles_local.sh.  PMC passes are separate runs with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_line.err
# the N > 1 launch structure (three graphs + eager high-priority sampling stream + RCCL all-reduces) with a forced 1-rank group: the per-GPU
# ceiling of the multi-GPU run
PAPC_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 50 --no-cpu-baseline > $O/bench_line_dist1.json 2> $O/bench_line_dist1.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o run -- python bench.py --no-cpu-baseline --no-padded-leg --steps 50 --warmup 10 > $O/prof_stats.log 2>&1
# one step's dispatch timeline (start / gap / duration per launch) with and without the sampling branch
# (under rocprofv3 the HOST falls behind the device, so the default structure's second graph -- the sampling pyramid on the side stream -- is
# launched late and lands at the end of the step: its trace is kept for the record, the dispatch-order reference is the in-graph fork)
timeout 400 rocprofv3 --kernel-trace -d $O/prof_tl -o run -- python bench.py --no-cpu-baseline --no-padded-leg --in-graph-fork --steps 50 > /dev/null 2>&1
python tools/step_timeline.py $O/prof_tl/run_results.db 40 > $O/timeline.txt 2>&1
timeout 400 rocprofv3 --kernel-trace -d $O/prof_tls -o run -- python bench.py --no-cpu-baseline --no-padded-leg --steps 50 > /dev/null 2>&1
python tools/step_timeline.py $O/prof_tls/run_results.db 40 > $O/timeline_side_graph.txt 2>&1
rm -rf $O/prof_tls
timeout 400 rocprofv3 --kernel-trace -d $O/prof_tlf -o run -- python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 50 > /dev/null 2>&1
python tools/step_timeline.py $O/prof_tlf/run_results.db 40 > $O/timeline_fixed_plan.txt 2>&1
rm -rf $O/prof_tl $O/prof_tlf
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py --no-cpu-baseline --no-graph --steps 6 --warmup 2 > $O/pmc_$c.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o run -- python bench.py --no-cpu-baseline --no-graph --steps 6 --warmup 2 > $O/pmc_sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $O/pmc_lds -o run -- python bench.py --no-cpu-baseline --no-graph --no-padded-leg --steps 6 --warmup 2 > $O/pmc_lds.log 2>&1
# the other BASELINE configs (bench.py --config): bench line + kernel stats; PFN also its HBM traffic
for c in msg_seg pfn basic; do
  timeout 400 python bench.py --config $c > $O/bench_line_$c.json 2> $O/bench_line_$c.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_$c -o run -- python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 5 > $O/prof_stats_$c.log 2>&1
done
# config 3 again with its radius branches in series (PAPC_MSG_STREAMS=0): per-kernel durations with one kernel on the device at a time
PAPC_MSG_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_msg_seg_serial -o run -- python bench.py --config msg_seg --no-cpu-baseline --steps 20 --warmup 5 > $O/prof_stats_msg_seg_serial.log 2>&1
for cfg in pfn msg_seg basic; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${cfg}_$c -o run -- python bench.py --config $cfg --no-cpu-baseline --no-graph --steps 6 --warmup 2 > $O/pmc_${cfg}_$c.log 2>&1
  done
done
find $O -name "*.csv" | head -40
tail -c 300 $O/bench_line.json
