# twelve-wave dX flavours (WN = 2): full GPU suite, then msg_seg / headline incl. the padded leg, PAPC_STREAM_NW12 = 1 | 0 on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/nw12; : > gpurun_out/nw12/out2.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/nw12/out2.txt
for i in 1 2 3; do
  for v in 1 0; do
    PAPC_STREAM_NW12=$v timeout 300 python bench.py --config msg_seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('msg_seg nw12=$v', d['value'], d['ms_per_step'])" >> gpurun_out/nw12/out2.txt
  done
done
for i in 1 2; do
  for v in 1 0; do
    PAPC_STREAM_NW12=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline nw12=$v', d['value'], d['ms_per_step'], d.get('value_padded'), d.get('ms_per_step_padded'))" >> gpurun_out/nw12/out2.txt
  done
done
cat gpurun_out/nw12/out2.txt
