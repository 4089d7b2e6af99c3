import os, sys
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import subprocess
for nb in ("2", "1"):
    os.environ["PAPC_GEMM_NBUF"] = nb
    print("NBUF", nb, flush=True)
    subprocess.run([sys.executable, "tools/ablate_gemm.py"])
