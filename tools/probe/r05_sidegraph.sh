O=gpurun_out/r05_sidegraph
mkdir -p $O
python -m pytest tests/test_gpu_step.py tests/test_gpu_bench.py -x -q 2>&1 | tail -4 > $O/pytest.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-padded-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side-graph', d['value'], d['ms_per_step'], d['config']['final_loss'])" >> $O/ab.txt 2>&1
python bench.py --no-cpu-baseline --no-padded-leg --in-graph-fork 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork      ', d['value'], d['ms_per_step'], d['config']['final_loss'])" >> $O/ab.txt 2>&1
done
for i in 1 2; do
python bench.py --config msg_seg --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('msg_seg side-graph', d['value'], d['ms_per_step'], d['config']['final_loss'])" >> $O/ab.txt 2>&1
python bench.py --config msg_seg --no-cpu-baseline --in-graph-fork 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('msg_seg fork      ', d['value'], d['ms_per_step'], d['config']['final_loss'])" >> $O/ab.txt 2>&1
done
python bench.py > $O/bench_full.log 2>&1
cat $O/pytest.log $O/ab.txt; tail -c 600 $O/bench_full.log
