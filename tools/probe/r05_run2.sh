mkdir -p gpurun_out/r05_run2
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_run2/pytest.log
python bench.py --no-cpu-baseline --no-padded-leg > gpurun_out/r05_run2/bench.log 2>&1
python bench.py --config pfn --no-cpu-baseline > gpurun_out/r05_run2/pfn.log 2>&1
PAPC_PFN_ZERO_PADDED=0 python bench.py --config pfn --no-cpu-baseline > gpurun_out/r05_run2/pfn_allrows.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r05_run2/prof_pfn -o run -- python /root/repo/bench.py --config pfn --no-cpu-baseline --steps 50 > /dev/null 2>&1
cd /root/repo
python tools/rocpd_summary.py gpurun_out/r05_run2/prof_pfn/run_results.db > gpurun_out/r05_run2/pfn_kstats.txt 2>&1
python tools/step_timeline.py gpurun_out/r05_run2/prof_pfn/run_results.db 40 > gpurun_out/r05_run2/pfn_timeline.txt 2>&1
rm -f gpurun_out/r05_run2/prof_pfn/*.db
for f in gpurun_out/r05_run2/*.log; do echo == $f; tail -c 700 $f; done
