"""Static check of the inline-asm operand ring of csrc/mlp_stream.hip (run by papc_amd.build on the `hipcc -S` output at every build).

Between a hidden ``global_load_dwordx4 v[a:b]`` (inside ;;#ASMSTART .. ;;#ASMEND) and the hand-placed ``s_waitcnt vmcnt(N)`` that covers it
no COMPILER instruction may read or write v[a:b]: a v_mov / v_accvgpr_write copy of a register whose load is still in flight copies stale
data, and the late data then lands in a register that has been given away.  Every basic block is walked linearly (a conservative
approximation: in-flight sets are carried across fall-through edges and dropped at a wait with a small enough count).

    python -m papc_amd._isa_audit <file.s> [kernel-substring]
"""
import re
import sys

_REG_RNG = re.compile(r"\bv\[(\d+):(\d+)\]")
_REG_ONE = re.compile(r"\bv(\d+)\b")


def _regs_of(text):
    out = set()
    for a, b in _REG_RNG.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in _REG_ONE.findall(text):
        out.add(int(a))
    return out


def audit(asm_text, want="stream_kernel"):
    """Returns {"kernels": names scanned, "loads": hidden loads seen, "violations": {kernel: [(line, load line, text, regs)]}}."""
    kern, in_asm, inflight, bad, nload, seen = None, False, [], {}, 0, []     # inflight: list of (set(regs), line no)
    for i, line in enumerate(asm_text.splitlines()):
        t = line.strip()
        m = re.match(r"^(_ZN4papc\w+):", t)
        if m:
            kern, inflight = (m.group(1) if want in m.group(1) and "Lb1E" in m.group(1) else None), []
            if kern:
                seen.append(kern)
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
            if t.startswith("global_load_dword"):     # (dwordx4 operand loads; dword loads of the compacted flavours' per-row weight / group)
                inflight.append((_regs_of(t.split(",")[0]), i + 1))
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
        touched = _regs_of(t.split(";")[0])
        for dst, ln in inflight:
            hit = touched & dst
            if hit:
                bad.setdefault(kern, []).append((i + 1, ln, t.split(";")[0].strip(), sorted(hit)))
    return {"kernels": seen, "loads": nload, "violations": bad}


def audit_ticket(asm_text):
    """csrc/pfn.hip's last-arriving-workgroup protocol (pfn_last_block): every ticket atomic (global_atomic_add) must sit behind an
    ``s_waitcnt vmcnt(0)`` with no global store between the wait and the atomic -- the partial rows a workgroup publishes have then been
    acknowledged before its ticket is taken (a workgroup-scope release fence alone does not emit that wait on gfx950).
    Returns {"atomics": n, "violations": [(line, text)]}."""
    lines = asm_text.splitlines()
    n, bad = 0, []
    for i, line in enumerate(lines):
        t = line.strip()
        if not t.startswith("global_atomic_add"):
            continue
        n += 1
        ok = False
        for j in range(i - 1, -1, -1):
            u = lines[j].strip()
            if re.match(r"^(_ZN|\.Lfunc_begin)", u) or u.startswith("s_endpgm"):
                break
            if u.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
                break
            if u.startswith("s_waitcnt") and "vmcnt(0)" in u:
                ok = True
                break
        if not ok:
            bad.append((i + 1, t))
    return {"atomics": n, "violations": bad}


def report(res):
    lines = ["scanned %d hidden loads in %d kernels" % (res["loads"], len(res["kernels"]))]
    for k, v in res["violations"].items():
        lines.append("%s: %d compiler instructions touch a register with a hidden load in flight" % (k, len(v)))
        for (ln, lo, txt, regs) in v[:6]:
            lines.append("   line %d (load at %d) %-60s v%s" % (ln, lo, txt[:60], regs))
    if not res["violations"]:
        lines.append("clean: no compiler instruction touches an in-flight asm-load destination")
    return "\n".join(lines)


if __name__ == "__main__":
    r = audit(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "stream_kernel")
    print(report(r))
    sys.exit(1 if r["violations"] else 0)
