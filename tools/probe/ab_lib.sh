# A/B two builds of the library on the same box: default vs papc_amd/libpapc_alt.so
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-175
  PAPC_LIB=$GRAFT_REPO_ROOT/papc_amd/libpapc_alt.so timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-175
done
