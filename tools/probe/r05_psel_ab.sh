set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/psel
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/psel/tests.txt
for i in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 >> gpurun_out/psel/on.txt
  PAPC_PSEL=0 timeout 200 python bench.py --no-cpu-baseline --no-padded-leg 2>/dev/null | tail -1 >> gpurun_out/psel/off.txt
done
timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 >> gpurun_out/psel/on_fixed.txt
PAPC_PSEL=0 timeout 200 python bench.py --no-cpu-baseline --no-padded-leg --diag-fixed-plan 2>/dev/null | tail -1 >> gpurun_out/psel/off_fixed.txt
cat gpurun_out/psel/tests.txt
python - <<'PY'
import json
for f in ("on","off","on_fixed","off_fixed"):
    for l in open(f"gpurun_out/psel/{f}.txt"):
        try:
            d=json.loads(l); print(f, d["value"], d["ms_per_step"])
        except Exception as e: print(f, "bad", l[:100])
PY
