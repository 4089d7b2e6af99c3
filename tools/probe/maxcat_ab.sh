#!/bin/bash
# A/B of the max-pooled layer's dX: stream DY_MAX (default) vs sparse-max on the tiled kernel vs sparse-max on the stream kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/mc
PAPC_SPARSE_MAX=1 timeout 900 python -m pytest tests/test_gpu_mlp.py -q -m gpu -x -k "sa_forward or group_all or msg_vs or stack_backward" > gpurun_out/mc/test.log 2>&1; echo "tests rc=$?" >> gpurun_out/mc/test.log
for i in 1 2; do
for v in "0 1" "1 0" "1 1"; do set -- $v
  PAPC_SPARSE_MAX=$1 PAPC_STREAM_MAXCAT=$2 python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sparse=$1 stream=$2', d['ms_per_step'], d['value'])" >> gpurun_out/mc/ab.log
done; done
for v in "0 1" "1 1"; do set -- $v
  d=/tmp/prof_$1; rm -rf $d
  PAPC_SPARSE_MAX=$1 PAPC_STREAM_MAXCAT=$2 rocprofv3 --kernel-trace --output-format csv -d $d -- python bench.py --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find $d -name '*kernel_trace.csv' | head -1)
  python tools/ktimeline.py $f > gpurun_out/mc/timeline_$1.txt 2>&1
done
