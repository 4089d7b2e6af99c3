# A/B the default library against papc_amd/libpapc_alt.so on one box (alternating runs)
O=gpurun_out/r05_ab_alt
mkdir -p $O
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-padded-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
PAPC_LIB=$GRAFT_REPO_ROOT/papc_amd/libpapc_alt.so python bench.py --no-cpu-baseline --no-padded-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alt ', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
done
python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base fixed', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
PAPC_LIB=$GRAFT_REPO_ROOT/papc_amd/libpapc_alt.so python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alt  fixed', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PAPC_LIB=$GRAFT_REPO_ROOT/papc_amd/libpapc_alt.so rocprofv3 --kernel-trace -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py $O/prof/run_results.db 40 > $O/timeline_alt_fixed.txt 2>&1
rm -rf $O/prof
cat $O/ab.txt
