mkdir -p gpurun_out/r05_run1
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_run1/pytest.log
python bench.py > gpurun_out/r05_run1/bench.log 2>&1
PAPC_DEFER_FOLDS=0 python bench.py --no-cpu-baseline --no-padded-leg > gpurun_out/r05_run1/bench_nodefer.log 2>&1
python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan > gpurun_out/r05_run1/bench_fixed.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r05_run1/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py gpurun_out/r05_run1/prof/run_results.db 40 > gpurun_out/r05_run1/timeline.txt 2>&1
rm -f gpurun_out/r05_run1/prof/*.db
for f in gpurun_out/r05_run1/*.log; do echo == $f; tail -c 1500 $f; done
