#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04h_tests.log 2>&1; tail -3 gpurun_out/r04h_tests.log | grep -v Docs
tools/r4_ab.sh r04h 2 "c3|" "c3_1t|--tune 15=16" "sh3|--emulate-shard 3/8 --no-calibrate" "sh3_1t|--emulate-shard 3/8 --no-calibrate --tune 15=16" "c5|--config C5 --steps 20" "c5_1t|--config C5 --steps 20 --tune 15=16"
