O=gpurun_out/r05_pfn_floor
mkdir -p $O
for P in 12000 6000 3000 1500 750 256 64; do
PAPC_BENCH_PFN_P=$P python bench.py --config pfn --no-cpu-baseline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$P', d['ms_per_step'], d['config'].get('families_ms_per_step'))" >> $O/floor.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for P in 12000 256; do
PAPC_BENCH_PFN_P=$P rocprofv3 --kernel-trace -d /root/repo/$O/prof$P -o run -- python /root/repo/bench.py --config pfn --no-cpu-baseline --steps 50 > /dev/null 2>&1
cd /root/repo; python tools/step_timeline.py $O/prof$P/run_results.db 30 > $O/timeline_P$P.txt 2>&1; rm -rf $O/prof$P; cd /tmp
done
cd /root/repo
cat $O/floor.txt $O/timeline_P12000.txt $O/timeline_P256.txt
