# timeline of one replayed PointNet-Basic step (config 0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/basictl
rm -rf /tmp/bt
timeout 300 rocprofv3 --kernel-trace -d /tmp/bt -o run -- python bench.py --config basic --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1 < /dev/null
db=$(find /tmp/bt -name '*.db' | head -1)
[ -n "$db" ] && python tools/step_timeline.py $db > gpurun_out/basictl/timeline.txt
cat gpurun_out/basictl/timeline.txt | cut -c1-120
