#!/bin/bash
# Reduced compile of mlp_stream.hip: keeps the dispatch line of ONE (amode, epi) pair and drops the generic STREAM_CASE / compacted flavours, so that
# hipcc builds two or three kernels in ~4 s instead of the whole file in ~3 min -- for register / LDS / scratch iterations on a new flavour
# (DESIGN 3.16).  Prints the resource usage of every kernel left.  Optional sed expressions are applied to the reduced copy first.
#   bash tools/probe/stream_one_flavour.sh "A_DY_DENSE && epi == EPI_STORE_RED" ['s/old/new/' ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KEEP="$1"; shift || true
OUT=${TMPDIR:-/tmp}/stream_one
mkdir -p "$OUT"
python - "$ROOT" "$KEEP" "$OUT/ms.hip" <<'PY'
import re, sys
root, keep, out = sys.argv[1:4]
s = open(root + "/papc_amd/csrc/mlp_stream.hip").read()
s = re.sub(r"\n\s*STREAM_CASE\([^\n]*", "", s)
s = s.replace("if (kb == 8 && p.Nout == 128) return stream_go<AMODE, EPI, 8, 4, true>(p, geo, st);", "")
s = s.replace("if (kb == 16 && p.Nout == 128) return stream_go<AMODE, EPI, 16, 2, true>(p, geo, st);", "")
lines = [l for l in s.split("\n") if not (l.strip().startswith("if (amode ==") and "return" in l and "stream_pick" in l and keep not in l)]
open(out, "w").write("\n".join(lines))
PY
for e in "$@"; do sed -i "$e" "$OUT/ms.hip"; done
/opt/rocm/bin/hipcc -S --cuda-device-only "$OUT/ms.hip" -o "$OUT/ms.s" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off \
    -I "$ROOT/include" -I "$ROOT/papc_amd/csrc" -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep "error\|Function Name\|VGPRs:\|ScratchSize\|LDS Size" | sed 's/.*remark: [^ ]* //; s/ \[-Rpass.*//' | paste - - - - | cut -c1-260
echo "(assembly: $OUT/ms.s)"
