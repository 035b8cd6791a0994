#!/bin/bash
# kernel timeline of one step (rocprofv3 --kernel-trace): tools/trace_step.sh <tag> [bench.py args...]  ->  gpurun_out/trace_<tag>.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu --no-emulated-shards --no-calibrate "$@" > $out/log.txt 2>&1
cd $root
python3 - "$out" > gpurun_out/trace_$tag.txt <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("csdev::", "")[:48], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# the last complete step of the timed loop: find the last k_gas_setup before the final profile runs... print the 6th from the end occurrence
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_gas_setup")]
# bench: warmup 2 + steps 6 timed + profile reps; take the step starting at the 6th setup (middle of the timed loop)
i0 = idx[5]; i1 = idx[6]
t0 = rows[i0][0]
for r in rows[i0:i1]:
    print("%8.1f %8.1f %7.1f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
print("step span %.1f us" % ((rows[i1][0] - t0) / 1e3))
PY
cat gpurun_out/trace_$tag.txt
