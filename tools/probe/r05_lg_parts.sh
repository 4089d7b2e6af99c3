# workgroups (= statistics partial rows) of the gather-add kernels: PAPC_LG_PARTS sweep on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lg; : > gpurun_out/lg/out.txt
for rep in 1 2; do
for v in 2048 1024 768 512 4096; do
  PAPC_LG_PARTS=$v timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed parts=$v', d['value'], d['ms_per_step'])" >> gpurun_out/lg/out.txt
done
done
cat gpurun_out/lg/out.txt
