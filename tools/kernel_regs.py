#!/usr/bin/env python3
"""Register, scratch and LDS use of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): tools/kernel_regs.py build/api.s [filter ...]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2:]
pat = re.compile(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", re.S)
rows = []
for m in pat.finditer(t):
    name = m.group(3)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("void csdev::", "")
    if flt and not any(f in name for f in flt):
        continue
    rows.append((name, int(m.group(6)), int(m.group(1)), int(m.group(5)), int(m.group(4)), int(m.group(2))))
for r in rows:
    print("%-60s vgpr %3d agpr %3d sgpr %3d scratch %5d lds %6d" % r)
