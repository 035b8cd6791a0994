#!/usr/bin/env python3
"""VALU instruction counts of the loops of one kernel in a gfx950 listing:  tools/isa_loops.py build/api.s '<mangled-name substring>'
A loop = a label that a later branch jumps back to; prints, per loop, the instructions between label and branch: VALU (v_*), of which
fp64 (v_*_f64), matrix (v_mfma*), scalar, LDS (ds_*), global/flat memory."""
import re
import sys

t = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(\S*%s[^\s:]*):" % re.escape(key), t, re.M)
if not m:
    sys.exit("kernel not found")
body = t[m.end():t.index(".end_amdhsa_kernel", m.end())]
lines = [l.strip() for l in body.split("\n")]
pos = {}
for i, l in enumerate(lines):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        pos[mm.group(1)] = i
print(m.group(1))
for i, l in enumerate(lines):
    mm = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", l)
    if mm:
        tgt = mm.group(1) or mm.group(2)
        if tgt in pos and pos[tgt] < i:
            seg = [x for x in lines[pos[tgt]:i] if x and not x.startswith((";", ".")) and not x.endswith(":")]
            valu = [x for x in seg if x.startswith("v_") and not x.startswith("v_nop")]
            print("loop %-12s lines %5d  VALU %4d  f64 %4d  mfma %3d  salu %4d  lds %3d  vmem %3d" % (
                tgt, len(seg), len(valu), len([x for x in valu if "_f64" in x]), len([x for x in valu if x.startswith("v_mfma")]),
                len([x for x in seg if x.startswith("s_")]), len([x for x in seg if x.startswith("ds_")]),
                len([x for x in seg if x.startswith(("global_", "flat_", "buffer_"))])))
