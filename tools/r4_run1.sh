#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_flux_fused.py tests/test_gpu_parity.py tests/test_gpu_c5.py tests/test_gpu_members.py -m gpu -q -x > gpurun_out/r04c_tests.log 2>&1; tail -12 gpurun_out/r04c_tests.log
tools/r4_ab.sh r04c 2 "c3|" "c3_off|--tune 15=1" "c3_on|--tune 15=2" "sh3|--emulate-shard 3/8" "sh3_off|--emulate-shard 3/8 --tune 15=1" "sh0|--emulate-shard 0/8" "sh7|--emulate-shard 7/8" "q1|--emulate-shard 1/4" "q1_off|--emulate-shard 1/4 --tune 15=1" "c5|--config C5 --steps 20" "c5_off|--config C5 --steps 20 --tune 15=1" "c2|--config C2" "c2_off|--config C2 --tune 15=1"
