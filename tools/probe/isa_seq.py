"""Load / wait / barrier / MFMA skeleton of one kernel in a hipcc -S file: L = global_load, Wn = s_waitcnt vmcnt(n), B = s_barrier, M = v_mfma,
| = basic-block label.  Shows at a glance whether a register prefetch ring survives (counted waits) or is drained (W0 at every merge).
    python tools/probe/isa_seq.py <file.s> <kernel-name-substring>"""
import itertools
import re
import sys

src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2]
start = [i for i, l in enumerate(src) if re.match(r"^_ZN4papc\w+:", l.strip()) and want in l][0]
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
seq = []
for l in src[start:end]:
    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
    if m:
        seq.append("W%s" % m.group(1))
    if "s_barrier" in l:
        seq.append("B")
    if "global_load" in l:
        seq.append("L")
    if re.match(r"^\.LBB\d+_\d+:", l.strip()):
        seq.append("|")
    if "v_mfma" in l:
        seq.append("M")
out = []
for k, g in itertools.groupby(seq):
    n = len(list(g))
    out.append(k if n == 1 else "%s*%d" % (k, n))
print(" ".join(out))
