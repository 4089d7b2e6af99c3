"""Builds libpapc_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python -m papc_amd.build [--force]

Each ``csrc/*.hip`` is compiled to an object (in parallel, skipped when newer than its sources and
headers) and linked into ``papc_amd/libpapc_hip.so``.  The index-exact kernels are compiled with
``-ffp-contract=off`` so the only FMAs are the ones the source spells out.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libpapc_hip.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function"]
# per-file extra flags
NOSLP = ["-fno-slp-vectorize"]
RES = ["-Rpass-analysis=kernel-resource-usage"]
EXTRA = {"mlp_stream.hip": NOSLP + RES, "mlp_gemm.hip": NOSLP, "mlp_dw.hip": NOSLP, "sampling.hip": NOSLP, "bn_ops.hip": NOSLP, "pfn.hip": NOSLP, "interp.hip": NOSLP, "nms.hip": NOSLP, "head.hip": NOSLP, "lingather.hip": NOSLP, "smallm.hip": NOSLP + RES}  # packed f32 VALU (v_pk_add_f32 ...) is slower than scalar beside MFMAs


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _check_no_spills(remarks):
    """mlp_stream.hip loads its operands with inline asm that hipcc's register allocator cannot see in flight: a spilled
    (or scratch-backed) register of such a kernel would be copied before its load has landed.  Refuse to build one."""
    import re
    name = None
    seen = set()
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name and "stream_kernel" in name and "Lb1E" in name:
            seen.add(name)
            if int(m.group(2)) != 0 and "SGPRs Spill" not in m.group(1):
                raise RuntimeError("mlp_stream.hip: %s has %s = %s (asm-loaded registers must not spill)" % (name, m.group(1), m.group(2)))
    if not seen:     # fail closed: a toolchain that words its remarks or mangles the names differently must not pass unchecked
        raise RuntimeError("mlp_stream.hip: no kernel-resource-usage remark of an asm-ring stream_kernel was found; the spill guard cannot vouch for this build")


def _audit_asm_ring(asm_path):
    """Static audit of mlp_stream.hip's hidden operand ring (papc_amd/_isa_audit.py): between an inline-asm global_load_dwordx4 and
    the hand-placed s_waitcnt that covers it no compiler instruction may touch the destination registers (a v_mov / AGPR copy of a
    register whose load is in flight copies stale data and the late data lands in a register that was given away).  Fails closed."""
    from ._isa_audit import audit, report
    res = audit(open(asm_path).read(), "stream_kernel")
    if res["violations"] or not res["kernels"] or res["loads"] == 0:
        raise RuntimeError("mlp_stream.hip: the asm-ring audit failed (%d kernels, %d hidden loads scanned)\n%s"
                           % (len(res["kernels"]), res["loads"], report(res)[-2000:]))


def _audit_pfn_tickets(cmd):
    """pfn.hip's fused tails: the ISA must wait for the published partial rows' acknowledgements before a ticket is taken
    (papc_amd/_isa_audit.py::audit_ticket).  Fails closed."""
    from ._isa_audit import audit_ticket
    asm = os.path.join(OBJ, "pfn.s")
    cmd_s = list(cmd)
    cmd_s[cmd_s.index("-c")] = "-S"
    cmd_s[cmd_s.index("-o") + 1] = asm
    r2 = subprocess.run(cmd_s + ["--cuda-device-only"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r2.returncode != 0:
        raise RuntimeError("hipcc -S failed on pfn.hip\n" + r2.stdout[-2000:])
    res = audit_ticket(open(asm).read())
    os.remove(asm)
    if res["violations"] or res["atomics"] == 0:
        raise RuntimeError("pfn.hip: the ticket audit failed (%d ticket atomics scanned): %s" % (res["atomics"], res["violations"][:4]))


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + [__file__]
    hdr_time = _newest(hdrs)
    extra_env = os.environ.get("PAPC_BUILD_DEFS", "").split()   # development aid: e.g. PAPC_BUILD_DEFS="-DPAPC_STREAM_WAIT0"
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            cmd = [_hipcc(), "-c", s, "-o", o] + COMMON + EXTRA.get(os.path.basename(s), []) + extra_env
            jobs.append((s, cmd))

    def run(job):
        s, cmd = job
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return s, r.returncode, r.stdout

    jobs_by_src = {s_: c_ for s_, c_ in jobs}
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, rc, out in ex.map(run, jobs):
                if os.path.basename(s) == "mlp_stream.hip" and rc == 0:
                    _check_no_spills(out)
                    asm = os.path.join(OBJ, "mlp_stream.s")
                    cmd_s = [c for c in jobs_by_src[s] if c not in RES]
                    cmd_s[cmd_s.index("-c")] = "-S"
                    cmd_s[cmd_s.index("-o") + 1] = asm
                    r2 = subprocess.run(cmd_s + ["--cuda-device-only"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                    if r2.returncode != 0:
                        raise RuntimeError("hipcc -S failed on mlp_stream.hip\n" + r2.stdout[-2000:])
                    _audit_asm_ring(asm)
                    os.remove(asm)
                    out = "\n".join(l for l in out.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l)
                if os.path.basename(s) == "pfn.hip" and rc == 0:
                    _audit_pfn_tickets(jobs_by_src[s])
                if verbose and out.strip():
                    print(out)
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s\n%s" % (s, out))
                if verbose:
                    print("[papc_amd.build] compiled", os.path.basename(s))
    if force or jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed\n" + r.stdout)
        if verbose:
            print("[papc_amd.build] linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
