# the pyramid's start placed between the end of SA1's forward and a little past SA2's (gate at sa1 / sa2 + a spinning delay behind it), same box
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for cfg in "sa2 0" "sa1 100" "sa1 140" "sa1 170" "sa1 200" "sa2 30" "sa2 60"; do
    set -- $cfg
    python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-padded-leg --fork $1 --side-delay-us $2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('fork $1 delay $2', 'mean', d['ms_per_step'], 'median', d['ms_median'], 'min', d['ms_min'])"
  done
done
