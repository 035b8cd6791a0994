#!/bin/bash
# copy the summaries of a tools/run_final.sh run into profiles/ under the round's names:  tools/copy_profiles.sh <run tag> <round tag>
set -e
t=$1; r=$2; o=gpurun_out
test -f $o/${t}_bench_c3.json
for n in c3 c2 c5 c5_mixed lorentz doppler PHCO2 c3_nomatrix shard0 shard3 shard7; do cp $o/${t}_bench_$n.json profiles/${r}_bench_$n.json; done
cp $o/${t}_mode_t.json profiles/${r}_mode_t.json
cp $o/prof_$t/stats/runc/*kernel_stats.csv profiles/${r}_kernel_stats.csv
cp $o/prof_$t/pmc_fetch_summary.csv profiles/${r}_pmc_fetch_summary.csv
cp $o/prof_$t/pmc_write_summary.csv profiles/${r}_pmc_write_summary.csv
cp $o/prof_$t/pmc_traffic.json profiles/pmc_traffic.json
cp $o/prof_$t/pmc_traffic.json profiles/${r}_pmc_traffic.json
python3 - <<PY
import json, bench
d = json.load(open("profiles/pmc_traffic.json"))
print("stamp of the profile:", d["source_sha16"], " loaded sources:", bench.source_stamp())
for n in ["c3", "c3_nomatrix", "c2", "c5", "c5_mixed", "lorentz", "doppler", "PHCO2", "shard0", "shard3", "shard7"]:
    b = json.loads(open(f"profiles/${r}_bench_{n}.json").readline())
    print(n, "ms/step %.3f" % b["ms_per_step"], {k: round(v, 3) for k, v in b["roofline"]["kernel_ms"].items()})
PY
