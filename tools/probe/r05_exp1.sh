mkdir -p gpurun_out/r05_exp1
for f in start sa2 sa3 loss; do python bench.py --no-cpu-baseline --fork $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork $f', d['value'], d['ms_per_step'])" ; done > gpurun_out/r05_exp1/forks.txt 2>&1
python bench.py --no-cpu-baseline --diag-fixed-plan 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed-plan', d['value'], d['ms_per_step'])" >> gpurun_out/r05_exp1/forks.txt 2>&1
python bench.py --no-cpu-baseline --no-overlap 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-overlap', d['value'], d['ms_per_step'])" >> gpurun_out/r05_exp1/forks.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r05_exp1/prof_fixed -o fixed -- python /root/repo/bench.py --no-cpu-baseline --diag-fixed-plan --steps 50 > /dev/null 2>&1
cd /root/repo
ls gpurun_out/r05_exp1/prof_fixed | head
cat gpurun_out/r05_exp1/forks.txt
