O=gpurun_out/r05_run10
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest.log
for i in 1 2; do
for v in "PAPC_SAMPLE_ORDER=0" "PAPC_SAMPLE_ORDER=1" "PAPC_SAMPLE_ORDER=2" "PAPC_SIDE_PRIO=1" "PAPC_SIDE_PRIO=-1"; do
env $v python bench.py --no-cpu-baseline --no-padded-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
done
done
python bench.py --config pfn --no-cpu-baseline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pfn', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
python bench.py --config basic --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('basic', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PAPC_SAMPLE_ORDER=2 rocprofv3 --kernel-trace -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py $O/prof/run_results.db 40 > $O/timeline_order2.txt 2>&1
rm -rf $O/prof
cat $O/pytest.log $O/ab.txt
