#!/bin/bash
# copy the summaries of a tools/final.sh <tag> run from gpurun_out/ (scratch) into profiles/ (tracked):  tools/copy_profiles.sh <tag>
tag=${1:?usage: tools/copy_profiles.sh <tag>}
o=gpurun_out
for f in $o/${tag}_bench_*.json; do [ -s "$f" ] && cp $f profiles/; done
for f in mode_t multi_overlap c5_full_parity shard_balance; do [ -s $o/${tag}_$f.json ] && cp $o/${tag}_$f.json profiles/; done
[ -s $o/gray_kat.json ] && cp $o/gray_kat.json profiles/${tag}_gray_kat.json   # (written by tests/test_gpu_kat.py)
for t in $tag ${tag}c5; do
  d=$o/prof_$t
  [ -d $d ] || continue
  cp "$(ls -t $(find $d/stats -name "*kernel_stats.csv") | head -1)" profiles/${t}_kernel_stats.csv
  cp "$(ls -t $(find $d/stats_concurrent -name "*kernel_stats.csv") | head -1)" profiles/${t}_kernel_stats_concurrent.csv 2>/dev/null
  cp $d/pmc_fetch_summary.csv profiles/${t}_pmc_fetch_summary.csv
  cp $d/pmc_write_summary.csv profiles/${t}_pmc_write_summary.csv
  cp $d/pmc_traffic.json profiles/${t}_pmc_traffic.json
done
[ -s $o/prof_$tag/pmc_traffic.json ] && cp $o/prof_$tag/pmc_traffic.json profiles/pmc_traffic.json
[ -s $o/prof_${tag}c5/pmc_traffic.json ] && cp $o/prof_${tag}c5/pmc_traffic.json profiles/pmc_traffic_c5.json
for t in c3 sh3; do [ -s $o/trace_${tag}_$t.txt ] && cp $o/trace_${tag}_$t.txt profiles/${tag}_trace_$t.txt; done
[ -s $o/${tag}_gputests.log ] && tail -3 $o/${tag}_gputests.log > profiles/${tag}_gputests_tail.txt
ls profiles | grep "^${tag}" | wc -l
