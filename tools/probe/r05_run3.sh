mkdir -p gpurun_out/r05_run3
python -m pytest tests/test_gpu_step.py tests/test_gpu_bench.py tests/test_gpu_dist.py tests/test_gpu_folds.py tests/test_gpu_seg.py -x -q 2>&1 | tail -8 > gpurun_out/r05_run3/pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-padded-leg > gpurun_out/r05_run3/bench_$i.log 2>&1
PAPC_ADAM_IN_GRAPH=0 python bench.py --no-cpu-baseline --no-padded-leg > gpurun_out/r05_run3/bench_eageradam_$i.log 2>&1
done
python bench.py --config pfn --no-cpu-baseline > gpurun_out/r05_run3/pfn.log 2>&1
PAPC_ADAM_IN_GRAPH=0 python bench.py --config pfn --no-cpu-baseline > gpurun_out/r05_run3/pfn_eageradam.log 2>&1
python bench.py --config basic --no-cpu-baseline > gpurun_out/r05_run3/basic.log 2>&1
python bench.py --config msg_seg --no-cpu-baseline > gpurun_out/r05_run3/msg_seg.log 2>&1
for f in gpurun_out/r05_run3/*.log; do echo == $f; tail -c 400 $f; done
