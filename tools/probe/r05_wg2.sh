O=gpurun_out/r05_wg2
mkdir -p $O
for i in 1 2; do
for v in 0 1 2 3; do
PAPC_STREAM_WG2=$v python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wg2=$v fixed', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
done
done
PAPC_STREAM_WG2=3 python -m pytest tests/test_gpu_stream.py tests/test_gpu_mlp.py -x -q 2>&1 | tail -3 >> $O/ab.txt
cd /tmp && export TMPDIR=/tmp
PAPC_STREAM_WG2=3 rocprofv3 --kernel-trace -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py $O/prof/run_results.db 40 2>&1 | grep "stream_kernel" > $O/timeline_wg2.txt
rm -rf $O/prof
cat $O/ab.txt $O/timeline_wg2.txt
