#!/bin/bash
# A/B of the 128 -> 256 dW on the row-streaming hybrid (dw_rowsx_kernel, deep prefetch ring) vs the staged kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/rx
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > gpurun_out/rx/test.log 2>&1; echo "rc=$?" >> gpurun_out/rx/test.log
for i in 1 2; do for v in 0 1; do
  PAPC_DW_ROWSX=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rowsx=$v', d['ms_per_step'], d['value'], d['roofline']['ms_per_step'])" >> gpurun_out/rx/ab.log
done; done
d=/tmp/prof_rx; rm -rf $d
PAPC_DW_ROWSX=1 rocprofv3 --kernel-trace --output-format csv -d $d -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python tools/ktimeline.py $(find $d -name '*kernel_trace.csv' | head -1) > gpurun_out/rx/timeline.txt 2>&1
grep "dw_" gpurun_out/rx/timeline.txt
