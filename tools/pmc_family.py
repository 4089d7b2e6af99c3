"""Per kernel family: HBM traffic per launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; unit KB).
    python tools/pmc_family.py <fetch dir> <write dir> > profiles/rNN_bench_pmc_family.json
FETCH_SIZE is doubled (gfx950 under-reports wide streaming reads by 2x: MI355X_MICROARCH.md, HBM section; calibrated in this repo on
streaming kernels of known traffic).  Families are the library's profiling families (papc_prof_*, include/papc_hip.h)."""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    n = re.sub(r"^void ", "", name).replace("papc::", "")
    if n.startswith(("dw_ws_kernel", "dw_kernel", "dw_xyz_kernel", "dw_rows_kernel", "dw_rowsx_kernel", "lingather_bwd", "pg_fold_kernel",
                     "xyz_l1_bwd_kernel", "dw_rows_max_kernel", "dw_max_fold_kernel", "dw_max_finalize_kernel")):
        return "bwd_dw_gemm"
    m = re.match(r"pg_gemm_kernel<(\d+),", n)      # planes GEMM (smallm.hip): epilogue 1 / 2 = forward, 3 = dX (+ BN-backward sums), 0 = dW partials (and the last dX)
    if m:
        return {1: "mlp_gemm_fwd", 2: "mlp_gemm_fwd", 3: "bwd_dx_gemm"}.get(int(m.group(1)), "bwd_dw_gemm")
    if n.startswith("lingather_fwd_kernel"):
        return "mlp_gemm_fwd"
    m = re.match(r"(?:gemm|stream)_kernel<(\d+),", n)
    if m:
        return "mlp_gemm_fwd" if int(m.group(1)) in (0, 1, 2, 6) else "bwd_dx_gemm"     # (A_PLAIN, A_BNRELU, A_GROUP, A_XYZ: forward operands)
    if n.startswith("pfn_kernel"):
        return "pfn"
    if n.startswith("fps_kernel"):
        return "fps"
    if n.startswith("ball_query"):
        return "ball_query"
    return None


def collect(d):
    agg = defaultdict(lambda: [0, 0.0])
    with open(d + "/run_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if "adam_kernel" in r["Kernel_Name"] or "adam_dev_kernel" in r["Kernel_Name"]:
                agg["_steps"][0] += 1          # one optimizer launch per step: the step count of the profiled run
            fam = family(r["Kernel_Name"])
            if fam:
                agg[fam][0] += 1
                agg[fam][1] += float(r["Counter_Value"])
    return agg


fetch, write = collect(sys.argv[1]), collect(sys.argv[2])
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --no-cpu-baseline "
               "--no-graph --steps 6 --warmup 2`; bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / kernel launches; per step = / adam_kernel launches of the same run"}
steps = max(fetch.get("_steps", [0])[0], write.get("_steps", [0])[0], 1)
out["_steps_profiled"] = steps
for fam in sorted((set(fetch) | set(write)) - {"_steps"}):
    nf, kf = fetch.get(fam, [0, 0.0])
    nw, kw = write.get(fam, [0, 0.0])
    n = max(nf, nw, 1)
    out[fam] = {"launches_profiled": n, "fetch_MB_per_launch_x2": round(2 * kf * 1024 / n / 1e6, 2),
                "write_MB_per_launch": round(kw * 1024 / n / 1e6, 2),
                "traffic_MB_per_launch": round((2 * kf + kw) * 1024 / n / 1e6, 2),
                "traffic_MB_per_step": round((2 * kf + kw) * 1024 / steps / 1e6, 1)}    # (kernels per launch differ between families: bench.py divides this by ITS launches per step)
print(json.dumps(out, indent=1))
