#!/bin/bash
# copy the summaries of a tools/r4_final.sh run into profiles/
o=gpurun_out
for f in $o/r04_bench_*.json; do cp $f profiles/; done
cp $o/r04_mode_t.json $o/r04_multi_overlap.json $o/r04_c5_full_parity.json $o/r04_shard_balance.json profiles/ 2>/dev/null
cp $o/gray_kat.json profiles/r04_gray_kat.json 2>/dev/null
for tag in r04 r04c5; do
  d=$o/prof_$tag
  [ -d $d ] || continue
  cp "$(ls -t $(find $d/stats -name "*kernel_stats.csv") | head -1)" profiles/${tag}_kernel_stats.csv
  cp "$(ls -t $(find $d/stats_concurrent -name "*kernel_stats.csv") | head -1)" profiles/${tag}_kernel_stats_concurrent.csv 2>/dev/null
  cp $d/pmc_fetch_summary.csv profiles/${tag}_pmc_fetch_summary.csv
  cp $d/pmc_write_summary.csv profiles/${tag}_pmc_write_summary.csv
  cp $d/pmc_traffic.json profiles/${tag}_pmc_traffic.json
done
cp $o/prof_r04/pmc_traffic.json profiles/pmc_traffic.json
cp $o/prof_r04c5/pmc_traffic.json profiles/pmc_traffic_c5.json
cp $o/trace_r04_sh3.txt profiles/r04_trace_shard3.txt 2>/dev/null
tail -3 $o/r04_gputests.log > profiles/r04_gputests_tail.txt
ls profiles | grep r04 | wc -l
