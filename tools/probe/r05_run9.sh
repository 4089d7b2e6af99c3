O=gpurun_out/r05_run9
mkdir -p $O
python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -5 > $O/pytest_head.log
timeout 600 python -m pytest tests/test_gpu_sampling.py -x -q 2>&1 | tail -12 > $O/pytest_bq.log
for i in 1 2; do
for v in "" "PAPC_HEAD_CHAIN=0"; do
env PAPC_BQ_GRID=0 $v python bench.py --no-cpu-baseline --no-padded-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
env PAPC_BQ_GRID=0 $v python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v] fixed', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
done
done
cd /tmp && export TMPDIR=/tmp
PAPC_BQ_GRID=0 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py $O/prof/run_results.db 40 > $O/timeline.txt 2>&1
rm -f $O/prof*/*.db
cat $O/pytest_head.log $O/pytest_bq.log $O/ab.txt; grep head_chain $O/timeline.txt
