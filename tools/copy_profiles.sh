#!/bin/bash
# copy the summaries of a tools/r3_final.sh run into profiles/
set -e
o=gpurun_out
test -f $o/r03_bench_c3.json
for f in $o/r03_bench_*.json; do cp $f profiles/; done
cp $o/r03_mode_t.json profiles/ 2>/dev/null || true
cp $o/r03_multi_overlap.json profiles/ 2>/dev/null || true
cp $o/r03_ubench*.txt $o/r03_ubench.json profiles/
cp $o/gray_kat.json profiles/r03_gray_kat.json
cp "$(ls -t $(find $o/prof_r03/stats -name "*kernel_stats.csv") | head -1)" profiles/r03_kernel_stats.csv   # (gpurun_out/ accumulates earlier runs: the newest)
cp "$(ls -t $(find $o/prof_r03/stats_concurrent -name "*kernel_stats.csv") | head -1)" profiles/r03_kernel_stats_concurrent.csv 2>/dev/null || true
cp $o/prof_r03/pmc_fetch_summary.csv profiles/r03_pmc_fetch_summary.csv
cp $o/prof_r03/pmc_write_summary.csv profiles/r03_pmc_write_summary.csv
cp $o/prof_r03/pmc_traffic.json profiles/pmc_traffic.json
cp $o/prof_r03/pmc_traffic.json profiles/r03_pmc_traffic.json
tail -3 $o/r03_gputests.log > profiles/r03_gputests_tail.txt
