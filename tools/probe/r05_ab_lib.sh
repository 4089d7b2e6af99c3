# same-box A/B of two builds of the library: papc_amd/libpapc_hip.so (new) against papc_amd/libpapc_hip_base.so (PAPC_LIB); headline, fixed plan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab; : > gpurun_out/ab/out.txt
B=$GRAFT_REPO_ROOT/papc_amd/libpapc_hip_base.so
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
  PAPC_LIB=$B timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
done
timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new  fixed', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
PAPC_LIB=$B timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base fixed', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new  fixed', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
PAPC_LIB=$B timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base fixed', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
[ -n "$1" ] && for c in $1; do
  timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new  $c', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
  PAPC_LIB=$B timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base $c', d['value'], d['ms_per_step'])" >> gpurun_out/ab/out.txt
done
cat gpurun_out/ab/out.txt
