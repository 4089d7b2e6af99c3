# whole step against the FPS kernel's threads per cloud (PAPC_FPS_THREADS: SA1's 4096-point clouds; default 512), same box, interleaved
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-padded-leg "$@" 2>/dev/null | python -c "
import json,sys,os;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('fps threads', os.environ.get('PAPC_FPS_THREADS'), 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"; }
for i in 1 2; do
  for v in 512 256 1024; do
    export PAPC_FPS_THREADS=$v
    run
  done
done
