cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 400 rocprofv3 --kernel-trace -d $O/prof_tl -o run -- python bench.py --no-cpu-baseline --no-padded-leg --in-graph-fork --steps 50 > /dev/null 2>&1
python tools/step_timeline.py $O/prof_tl/run_results.db 40 > $O/timeline.txt 2>&1
rm -rf $O/prof_tl
tail -1 $O/timeline.txt
