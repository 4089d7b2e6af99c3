cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --lr ${LR:-1e-3} --dump-trajectory /tmp/$name.npz $FLAGS > /dev/null 2>&1; }
FLAGS="" run g_ov A=1
FLAGS="--no-overlap" run g_noov A=1
FLAGS="--no-graph" run e_ov A=1
FLAGS="--no-graph --no-overlap" run e_noov A=1
FLAGS="--no-graph --no-overlap" run e_noov2 A=1
FLAGS="" run g_ov_noplanes PAPC_PLANES=0
FLAGS="--no-graph --no-overlap" run e_noov_noplanes PAPC_PLANES=0
python - <<'P'
import numpy as np
names=["g_ov","g_noov","e_ov","e_noov","e_noov2","g_ov_noplanes","e_noov_noplanes"]
d={n:np.load("/tmp/%s.npz"%n) for n in names}
for n in names: print("%-16s"%n, " ".join("%.5f"%v for v in d[n]["loss"]))
ref=d["e_noov"]
for n in names: print(n, "max param diff vs e_noov %.3e"%np.max(np.abs(d[n]["params"]-ref["params"])))
P
