#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/r04w_tests.log 2>&1; tail -5 $out/r04w_tests.log
tools/r4_ab.sh r04w 2 "c3|" "c3off|--tune 16=1" "half|--emulate-shard 1/2 --no-calibrate" "shard|--emulate-shard 3/8 --no-calibrate" "c5|--config C5 --steps 20" "c5off|--config C5 --steps 20 --tune 16=1"
