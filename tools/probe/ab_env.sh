# A/B an environment switch on the same box:  bash tools/probe/ab_env.sh VAR=value
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-175
  env "$@" timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-175
done
