# the max layer's dX with a gathered sparse half (SPX): parity tests of the path, then same-box A/B against the [P | A] matrix form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/spx; : > gpurun_out/spx/out.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "max_layer or cabi or stack or step or model or config2 or bench" 2>&1 | tail -4 >> gpurun_out/spx/out.txt
for i in 1 2 3; do
  for v in 1 0; do
    PAPC_STREAM_SPX=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spx=$v', d['value'], d['ms_per_step'])" >> gpurun_out/spx/out.txt
  done
done
for v in 1 0; do
  PAPC_STREAM_SPX=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed spx=$v', d['value'], d['ms_per_step'])" >> gpurun_out/spx/out.txt
done
cat gpurun_out/spx/out.txt
