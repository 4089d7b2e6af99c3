O=gpurun_out/r05_run5
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python bench.py > $O/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/rocpd_summary.py $O/prof/run_results.db > $O/kstats.txt 2>&1
python tools/step_timeline.py $O/prof/run_results.db 40 > $O/timeline.txt 2>&1
rm -f $O/prof/*.db
for f in $O/*.log; do echo == $f; tail -c 1500 $f; done
