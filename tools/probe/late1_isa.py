"""Static check of the inline-asm operand ring of mlp_stream.hip: between a hidden `global_load_dwordx4 v[a:b]` (inside ;;#ASMSTART ..
;;#ASMEND) and the hand-placed `s_waitcnt vmcnt(N)` that covers it, no COMPILER instruction may read or write v[a:b] (a v_mov / v_accvgpr_write
copy of a register whose load is still in flight copies stale data, and the late data then lands in a register that has been given away).
    python tools/probe/late1_isa.py <file.s> [kernel-substring]
Walks every basic block linearly (a conservative approximation: in-flight sets are carried across fall-through edges and dropped at a wait
with a small enough count)."""
import re
import sys

src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else "stream_kernel"
reg_rng = re.compile(r"\bv\[(\d+):(\d+)\]")
reg_one = re.compile(r"\bv(\d+)\b")


def regs_of(text):
    out = set()
    for a, b in reg_rng.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in reg_one.findall(text):
        out.add(int(a))
    return out


kern, in_asm, inflight, bad, nload = None, False, [], {}, 0     # inflight: list of (set(regs), line no)
for i, line in enumerate(src):
    t = line.strip()
    m = re.match(r"^(_ZN4papc\w+):", t)
    if m:
        kern, inflight = (m.group(1) if want in m.group(1) and "Lb1E" in m.group(1) else None), []
        continue
    if kern is None:
        continue
    if t.startswith("s_endpgm"):
        kern = None
        continue
    if ";;#ASMSTART" in t:
        in_asm = True
        continue
    if ";;#ASMEND" in t:
        in_asm = False
        continue
    if in_asm:
        if t.startswith("global_load_dwordx4"):
            dst = regs_of(t.split(",")[0])
            inflight.append((dst, i + 1))
            nload += 1
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight     # loads return in order: all but the youngest n landed
            if n == 0:
                inflight = []
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
        inflight = []
        continue
    touched = regs_of(t.split(";")[0])
    for dst, ln in inflight:
        hit = touched & dst
        if hit:
            bad.setdefault(kern, []).append((i + 1, ln, t.split(";")[0].strip(), sorted(hit)))
print("scanned %d hidden loads" % nload)
for k, v in bad.items():
    print("%s: %d compiler instructions touch a register with a hidden load in flight" % (k, len(v)))
    for (ln, lo, txt, regs) in v[:6]:
        print("   line %d (load at %d) %-60s v%s" % (ln, lo, txt[:60], regs))
if not bad:
    print("clean: no compiler instruction touches an in-flight asm-load destination")
sys.exit(1 if bad else 0)
