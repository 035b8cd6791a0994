#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel for a library under build/:  tools/pmc_ab.sh lib.so [bench args]   (separate --pmc passes)
lib=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_${lib%.so}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export CLEARSKY_HIP_LIB=$root/build/$lib
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $out/$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("csdev::", "")]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:9]:
        print(f"{name} {k}: {n} launches, {v / n / 1024:.1f} MB per launch")
PY
