#!/bin/bash
# Profiling recipe (run on the GPU box from the repo root):  tools/profile.sh <tag> [bench.py args...]
#   pass 1: rocprofv3 --kernel-trace --stats           -> gpurun_out/prof_<tag>/stats/*kernel_stats.csv
#   pass 2, 3: --pmc FETCH_SIZE / --pmc WRITE_SIZE      -> gpurun_out/prof_<tag>/{fetch,write}/*counter_collection.csv
# (counters in their own passes, the program directly after `--`; summaries are copied into profiles/ by hand)
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
rm -rf $out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# (--tune 2=0,7=0: kernels one after the other on one stream, as in the HIP-event profile bench.py's roofline is built from; with the side
#  streams of the default step a kernel's trace duration includes the time it shares the chip with the kernels beside it -- that trace
#  is kept too, as stats_concurrent)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu --no-emulated-shards --tune 2=0,7=0 "$@" > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_concurrent -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu --no-emulated-shards "$@" > $out/stats_concurrent.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-emulated-shards "$@" > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu --no-emulated-shards "$@" > $out/write.log 2>&1
cd $root
python3 - "$out" "$tag" "$@" <<'PY'
import csv, glob, json, sys, collections
sys.path.insert(0, ".")
import bench
out, tag, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
traffic = dict(tag=tag, source_sha16=bench.source_stamp(), config=(bargs[bargs.index("--config") + 1] if "--config" in bargs else "C3"),
               bench_args=bargs, units="KB per dispatch (mean over dispatches); FETCH_SIZE and WRITE_SIZE in separate passes",
               calibration=("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, KB per dispatch as reported.  MI355X_MICROARCH.md (HBM): "
                            "FETCH_SIZE counts fabric read requests (Infinity-Cache hits included) and reports half the bytes of 16-B-per-lane "
                            "streaming reads; the reads here are 8 B per lane (sigma, nu), 32-B scalar record loads and 32-B per-lane record "
                            "loads -- widths the guide leaves uncalibrated -- so the figure is a floor, at most 2x low; WRITE_SIZE is exact for "
                            "streaming stores"), kernels={})
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
for name in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"].split("(")[0]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    with open(f"{out}/pmc_{name}_summary.csv", "w") as fo:
        fo.write("kernel,launches,avg_KB_per_launch\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"\"{k}\",{n},{v / n:.1f}\n")
            if k.startswith(("void csdev::", "csdev::")):
                kk = k.replace("void ", "").replace("csdev::", "")
                traffic["kernels"].setdefault(kk, {})[("FETCH" if name == "fetch" else "WRITE") + "_SIZE_KB"] = round(v / n, 1)
    print(open(f"{out}/pmc_{name}_summary.csv").read())
# dominant kernel = the one with the largest total time in the kernel trace
best = ("", 0.0)
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdev::" in r["Name"] and float(r["TotalDurationNs"]) > best[1]:
            best = (r["Name"].split("(")[0].replace("void ", "").replace("csdev::", ""), float(r["TotalDurationNs"]))
traffic["dominant"] = best[0]
# bytes per step: every kernel of the step once per launch group (the bench column has ONE merged group), the two near tiers separately
step = [k for k in traffic["kernels"] if not k.startswith(("k_transpose", "k_cheb_setup", "k_cascade_setup", "k_devfn", "k_faddeeva", "k_fill", "k_table_log"))]
traffic["bytes_per_step"] = sum((traffic["kernels"][k].get("FETCH_SIZE_KB", 0.0) + traffic["kernels"][k].get("WRITE_SIZE_KB", 0.0)) * 1024.0 for k in step)
traffic["bytes_per_step_kernels"] = step
json.dump(traffic, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic)[:600])
PY
