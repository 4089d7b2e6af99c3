# per-launch duration of the compacted max-layer dX kernel with and without the streamed psel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/psel; : > gpurun_out/psel/kern.txt
for v in 1 0; do
  rm -rf /tmp/pp$v
  PAPC_PSEL=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$v -o p -- python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 60 > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/pp$v -name '*kernel_stats.csv' | head -1)
  echo "PAPC_PSEL=$v $f" >> gpurun_out/psel/kern.txt
  [ -n "$f" ] && grep -E 'stream_kernel<4, *2' "$f" >> gpurun_out/psel/kern.txt < /dev/null
done
cat gpurun_out/psel/kern.txt
