#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_sq2
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/a -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --tune 2=0,7=0 > $out/a.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq2/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void csdev::", "").replace("csdev::", "")
        res[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(res.items()):
    if not n.startswith("k_"): continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(n[:34].ljust(34), " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(m.items())), " lanes/instr=%.1f" % (m["SQ_THREAD_CYCLES_VALU"] / max(m["SQ_ACTIVE_INST_VALU"], 1)))
PY
