# config 3 (MSG segmentation) with and without the gather-add backward that reads dz alone (PAPC_LG_PP), same box, interleaved
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for v in 0 1; do
    PAPC_LG_PP=$v python bench.py --config msg_seg --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('pp $v msg_seg ms', d['ms_per_step'], 'value', d['value'])"
  done
done
