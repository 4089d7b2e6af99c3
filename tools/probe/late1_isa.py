"""Command-line front of papc_amd._isa_audit (the build's audit of mlp_stream.hip's inline-asm operand ring).
    python tools/probe/late1_isa.py <file.s> [kernel-substring]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from papc_amd._isa_audit import audit, report  # noqa: E402

r = audit(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "stream_kernel")
print(report(r))
sys.exit(1 if r["violations"] else 0)
