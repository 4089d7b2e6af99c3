"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file) into one line per kernel.
    python tools/kernel_resources.py /tmp/res.txt [filter-substring]"""
import re, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r'remark: Function Name: ', t)[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    g = lambda pat: int(re.search(pat + r': (\d+)', b).group(1))
    rows.append((name, g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
for r in sorted(rows, key=lambda r: -r[1]):
    if flt in r[0]:
        print("%-90s vgpr %3d agpr %3d spill %3d scratch %4d occ %d lds %6d" % (r[0][:90], *r[1:]))
