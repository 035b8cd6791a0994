#!/bin/bash
# SQ counters per kernel of one step (tools/pmc_sq.sh [bench.py args, e.g. --config C5]), kernels in serial order (--tune 2=0,7=0), two passes of four counters; summaries -> gpurun_out/pmc_sq_<n>.csv
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_sq
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/a -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-emulated-shards --tune 2=0,7=0 "$@" > $out/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/b -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-emulated-shards --tune 2=0,7=0 "$@" > $out/b.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void csdev::", "").replace("csdev::", "")
        res[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, d in sorted(res.items()):
    if not n.startswith("k_"): continue
    out[n] = {c: sum(v) / len(v) for c, v in d.items()}
json.dump(out, open("gpurun_out/pmc_sq/summary.json", "w"), indent=1)
for n, d in out.items():
    print(n[:34].ljust(34), " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(d.items())))
PY
