#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
tools/r4_ab.sh r04k 2 "c3|" "c3_scan|--tune 15=2" "h1|--emulate-shard 1/2 --no-calibrate" "h1_scan|--emulate-shard 1/2 --no-calibrate --tune 15=2" "q1|--emulate-shard 1/4 --no-calibrate" "q1_off|--emulate-shard 1/4 --no-calibrate --tune 15=1"
