#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04i_tests.log 2>&1; tail -3 gpurun_out/r04i_tests.log | grep -v Docs
tools/r4_ab.sh r04i 2 "sh3|--emulate-shard 3/8 --no-calibrate" "sh3_st|--emulate-shard 3/8 --no-calibrate --tune 15=64" "sh3_off|--emulate-shard 3/8 --no-calibrate --tune 15=1" "c2|--config C2" "c2_st|--config C2 --tune 15=64" "c2_off|--config C2 --tune 15=1" "q1|--emulate-shard 1/4 --no-calibrate" "q1_off|--emulate-shard 1/4 --no-calibrate --tune 15=1" "sh0|--emulate-shard 0/8 --no-calibrate" "sh7|--emulate-shard 7/8 --no-calibrate"
