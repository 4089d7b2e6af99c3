# where the forced-1-rank N > 1 launch structure (two graphs + RCCL all-reduces + eager Adam) spends its extra 0.1 ms per step: dispatch timeline of
# one step under rocprofv3 (kernel trace; the host falls behind under the profiler, so gaps are upper bounds), next to the un-profiled step times
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp PAPC_FORCE_DIST=1
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('dist1 $*', d['ms_per_step'], d.get('ms_median'), d.get('ms_min'))"; }
run
run --eager-sampling
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-padded-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('n=1 structure', d['ms_per_step'], d['ms_median'], d['ms_min'])"
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/d1tl -o run -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline >/dev/null 2>&1)
db=$(find /tmp/d1tl -name "*results.db" | head -1)
python tools/step_timeline.py "$db" 20 2>&1 | awk '$3+0 > 3.0 || /step span/ || /all_reduce|ccl|adam|fold_jobs|flag_/' | head -60
