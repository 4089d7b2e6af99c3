PAPC_SPARSE_MAX=1 timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_models.py -x -q 2>&1 | tail -8
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-180; PAPC_SPARSE_MAX=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-180; done
