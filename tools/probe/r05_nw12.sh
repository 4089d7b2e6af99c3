# three waves per SIMD for the XYZ_RED dX (stream_kernel<..., 12>): tests of the touched path + same-box A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/nw12; : > gpurun_out/nw12/out.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "xyz or stack or cabi or step or model or bench" 2>&1 | tail -3 >> gpurun_out/nw12/out.txt
for i in 1 2 3; do
  for v in 1 0; do
    PAPC_STREAM_NW12=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nw12=$v', d['value'], d['ms_per_step'])" >> gpurun_out/nw12/out.txt
  done
done
for v in 1 0; do
  PAPC_STREAM_NW12=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed nw12=$v', d['value'], d['ms_per_step'])" >> gpurun_out/nw12/out.txt
done
cat gpurun_out/nw12/out.txt
