#!/bin/bash
# A/B of the max-pooled layer without its stored output (PAPC_NOSTORE) on the headline step + per-kernel timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/ns
timeout 900 python -m pytest tests/test_gpu_mlp.py -q -m gpu -x -k "without_stored or stack_backward" > gpurun_out/ns/test.log 2>&1; echo "tests rc=$?" >> gpurun_out/ns/test.log
for i in 1 2; do
for v in 0 1; do
  PAPC_NOSTORE=$v python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nostore=$v', d['ms_per_step'], d['value'])" >> gpurun_out/ns/ab.log
done; done
for v in 1; do
  d=/tmp/prof_$v; rm -rf $d
  PAPC_NOSTORE=$v rocprofv3 --kernel-trace --output-format csv -d $d -- python bench.py --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find $d -name '*kernel_trace.csv' | head -1)
  python tools/ktimeline.py $f > gpurun_out/ns/timeline_$v.txt 2>&1
done
