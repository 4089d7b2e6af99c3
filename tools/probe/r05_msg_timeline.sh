# timeline of one replayed msg_seg step (streams mode): where the device idles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/msgtl
rm -rf /tmp/mt
timeout 400 rocprofv3 --kernel-trace -d /tmp/mt -o run -- python bench.py --config msg_seg --no-cpu-baseline --steps 12 --warmup 4 > /dev/null 2>&1 < /dev/null
db=$(find /tmp/mt -name '*.db' | head -1)
[ -n "$db" ] && python tools/step_timeline.py $db > gpurun_out/msgtl/timeline.txt
tail -1 gpurun_out/msgtl/timeline.txt
