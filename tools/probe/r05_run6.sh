O=gpurun_out/r05_run6
mkdir -p $O
for f in start sa2 sa3 loss; do python bench.py --no-cpu-baseline --no-padded-leg --fork $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork $f', d['value'], d['ms_per_step'])" ; done > $O/forks.txt 2>&1
python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed-plan', d['value'], d['ms_per_step'])" >> $O/forks.txt 2>&1
python bench.py --no-cpu-baseline --no-padded-leg --no-overlap 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-overlap', d['value'], d['ms_per_step'])" >> $O/forks.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --steps 50 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_fixed -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py $O/prof/run_results.db 40 > $O/timeline.txt 2>&1
python tools/step_timeline.py $O/prof_fixed/run_results.db 40 > $O/timeline_fixed.txt 2>&1
python tools/rocpd_summary.py $O/prof_fixed/run_results.db > $O/kstats_fixed.txt 2>&1
rm -f $O/prof*/*.db
cat $O/forks.txt
