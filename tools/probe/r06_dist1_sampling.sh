# forced-1-rank N > 1 structure: the pyramid as a gated hipGraph on the side stream (default) against eager launches on a high-priority side stream
# (--eager-sampling), same box, interleaved
cd "$GRAFT_REPO_ROOT"
export PAPC_FORCE_DIST=1
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline ${@:2} 2>/dev/null | grep "^{" | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('dist1 ${*:2}', d['ms_per_step'], d.get('ms_median'), d.get('ms_min'))"; }
p=29540
for i in 1 2 3; do
  run $p; p=$((p+1))
  run $p --eager-sampling; p=$((p+1))
done
