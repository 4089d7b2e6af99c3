#!/bin/bash
# SQ counters of the dW kernels (one pass per counter group; --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcdw; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p1 -o run -- python bench.py --no-cpu-baseline --no-graph --steps 4 --warmup 1 > $O/p1.log 2>&1
python tools/pmc_table.py /tmp/p1 60 > $O/t1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS --output-format csv -d /tmp/p2 -o run -- python bench.py --no-cpu-baseline --no-graph --steps 4 --warmup 1 > $O/p2.log 2>&1
python tools/pmc_table.py /tmp/p2 60 > $O/t2.txt 2>&1
grep "kernel\|dw_\|stream_kernel" $O/t1.txt | cut -c1-250
grep "kernel\|dw_\|stream_kernel" $O/t2.txt | cut -c1-250
