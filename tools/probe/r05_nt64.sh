O=gpurun_out/r05_nt64
mkdir -p $O
for i in 1 2 3; do
for v in 0 64; do
PAPC_STREAM_NT=$v python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt=$v fixed', d['value'], d['ms_per_step'])" >> $O/ab.txt 2>&1
done
done
PAPC_STREAM_NT=64 python -m pytest tests/test_gpu_stream.py tests/test_gpu_mlp.py tests/test_gpu_compact.py -x -q 2>&1 | tail -3 >> $O/ab.txt
cd /tmp && export TMPDIR=/tmp
PAPC_STREAM_NT=64 rocprofv3 --kernel-trace -d /root/repo/$O/prof -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/step_timeline.py $O/prof/run_results.db 40 2>&1 | grep "stream_kernel" > $O/timeline.txt
rm -rf $O/prof
cat $O/ab.txt $O/timeline.txt
