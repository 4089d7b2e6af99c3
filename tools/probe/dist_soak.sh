cd $GRAFT_REPO_ROOT
ok=0; bad=0
for i in $(seq 1 24); do
  port=$((29600 + i))
  extra="--dry-run"; [ $((i % 3)) = 0 ] && extra="--steps 20"
  if PAPC_FORCE_DIST=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --require-graph $extra > /tmp/soak_$i.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); tail -5 /tmp/soak_$i.log; fi
done
echo "soak: ok=$ok bad=$bad"
