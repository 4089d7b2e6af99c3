# one extremum per channel in the fused group max (sign(gamma) hint): full GPU suite, then PAPC_GSIGN = 1 | 0 on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gsign; : > gpurun_out/gsign/out.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> gpurun_out/gsign/out.txt
for i in 1 2 3; do
  for v in 1 0; do
    PAPC_GSIGN=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gsign=$v', d['value'], d['ms_per_step'])" >> gpurun_out/gsign/out.txt
  done
done
for v in 1 0 1 0; do
  PAPC_GSIGN=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed gsign=$v', d['value'], d['ms_per_step'])" >> gpurun_out/gsign/out.txt
done
for v in 1 0; do
  PAPC_GSIGN=$v timeout 300 python bench.py --config msg_seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('msg_seg gsign=$v', d['value'], d['ms_per_step'])" >> gpurun_out/gsign/out.txt
done
cat gpurun_out/gsign/out.txt
