# PAPC_POINT_LISTS=1 (lists for compacted stacks only) against =2 (padded stacks too, padding copies collapsed): config 3 and the headline, interleaved
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for v in 1 2; do
    PAPC_POINT_LISTS=$v python bench.py --config msg_seg --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('lists=$v msg_seg ms', d['ms_per_step'])"
    PAPC_POINT_LISTS=$v python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('lists=$v ssg ms', d['ms_per_step'], 'padded', d['ms_per_step_padded'])"
  done
done
