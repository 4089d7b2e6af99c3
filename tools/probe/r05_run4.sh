mkdir -p gpurun_out/r05_run4
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05_run4/pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-padded-leg > gpurun_out/r05_run4/bench_$i.log 2>&1
PAPC_WSTATS=0 python bench.py --no-cpu-baseline --no-padded-leg > gpurun_out/r05_run4/bench_corr_$i.log 2>&1
done
for f in gpurun_out/r05_run4/*.log; do echo == $f; tail -c 300 $f; done
