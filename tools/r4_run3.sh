#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_flux_fused.py tests/test_gpu_c5.py -m gpu -q -x > gpurun_out/r04e_tests.log 2>&1; tail -3 gpurun_out/r04e_tests.log
tools/r4_ab.sh r04e 2 "c5|--config C5 --steps 20" "c5_off|--config C5 --steps 20 --tune 15=1" "c5_3w|--config C5 --steps 20 --tune 15=8" "sh3|--emulate-shard 3/8"
