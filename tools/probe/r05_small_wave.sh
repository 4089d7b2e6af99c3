# one-wave finalizes: headline and msg_seg, tests of the touched paths
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sw; : > gpurun_out/sw/out.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "bn or stack or step or seg or model or cabi or compact" 2>&1 | tail -3 >> gpurun_out/sw/out.txt
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'])" >> gpurun_out/sw/out.txt
  timeout 300 python bench.py --config msg_seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('msg_seg', d['value'], d['ms_per_step'])" >> gpurun_out/sw/out.txt
done
timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed', d['value'], d['ms_per_step'])" >> gpurun_out/sw/out.txt
timeout 200 python bench.py --config basic --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('basic', d['value'], d['ms_per_step'])" >> gpurun_out/sw/out.txt
cat gpurun_out/sw/out.txt
