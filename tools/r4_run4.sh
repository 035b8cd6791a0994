#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04f_tests.log 2>&1; tail -3 gpurun_out/r04f_tests.log
tools/r4_ab.sh r04f 2 "c3|" "c3_sep|--tune 15=16" "sh3|--emulate-shard 3/8" "sh3_sep|--emulate-shard 3/8 --tune 15=16" "sh0|--emulate-shard 0/8" "sh7|--emulate-shard 7/8" "c5|--config C5 --steps 20" "c5_sep|--config C5 --steps 20 --tune 15=16" "c2|--config C2" "c2_sep|--config C2 --tune 15=16"
