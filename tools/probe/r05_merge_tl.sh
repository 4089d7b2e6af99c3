O=gpurun_out/r05_merge
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
PAPC_HEAD_MERGE=$m rocprofv3 --kernel-trace -d /root/repo/$O/prof$m -o run -- python /root/repo/bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan --steps 50 > /dev/null 2>&1
cd /root/repo; python tools/step_timeline.py $O/prof$m/run_results.db 40 2>&1 | grep -E "head_|softmax|pg_final|pg_prep_kernel<4>|step span" > $O/tl$m.txt; rm -rf $O/prof$m; cd /tmp
done
cd /root/repo; echo merge=1; cat $O/tl1.txt; echo merge=0; cat $O/tl0.txt
