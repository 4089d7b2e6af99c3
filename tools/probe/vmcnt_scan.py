"""Every COMPILER-placed `s_waitcnt vmcnt(N)` (outside ;;#ASMSTART .. ;;#ASMEND) of the inline-asm instantiations of mlp_stream.hip, per
kernel: {N: occurrences}.  DESIGN.md 3.8a: the compiler counts only the loads / stores it knows of; these are the waits a hidden load
younger than a compiler-visible one would break.
    hipcc -S papc_amd/csrc/mlp_stream.hip -o /tmp/ms.s --offload-arch=gfx950 -O3 -std=c++17 -I include -I papc_amd/csrc --cuda-device-only ...
    python tools/probe/vmcnt_scan.py /tmp/ms.s"""
import collections
import re
import sys

kern, in_asm, res = None, False, collections.defaultdict(collections.Counter)
for line in open(sys.argv[1]):
    t = line.strip()
    m = re.match(r"^(_ZN4papc13stream_kernel\w+):", t)
    if m:
        kern = m.group(1) if "Lb1E" in m.group(1) else None
        continue
    if kern is None:
        continue
    if t.startswith("s_endpgm"):
        kern = None
    elif ";;#ASMSTART" in t:
        in_asm = True
    elif ";;#ASMEND" in t:
        in_asm = False
    elif not in_asm:
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            res[kern][int(m.group(1))] += 1
for k, v in res.items():
    print(k, dict(sorted(v.items())))
