# the sampling stream under a CU mask (hipExtStreamCreateWithCUMask): step time, and whether the mask survives the graph replay (ball query duration)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/cumask; : > gpurun_out/cumask/out.txt
for m in 0 128 64 32 0 32; do
  timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --side-cu-mask $m 2>gpurun_out/cumask/err_$m.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mask $m', d['value'], d['ms_per_step'])" >> gpurun_out/cumask/out.txt
done
for m in 0 32; do
  rm -rf /tmp/cm$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cm$m -o p -- python bench.py --no-cpu-baseline --no-padded-leg --side-cu-mask $m --steps 40 > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/cm$m -name '*kernel_stats.csv' | head -1)
  echo "rocprofv3, mask $m:" >> gpurun_out/cumask/out.txt
  [ -n "$f" ] && grep -E 'ball_query|fps_kernel<512' "$f" | cut -d, -f1-4 | cut -c1-120 >> gpurun_out/cumask/out.txt < /dev/null
done
cat gpurun_out/cumask/out.txt; tail -3 gpurun_out/cumask/err_32.txt
