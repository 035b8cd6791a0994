#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_flux_fused.py tests/test_gpu_interp.py tests/test_gpu_c5.py -m gpu -q -x > gpurun_out/r04l_tests.log 2>&1; tail -3 gpurun_out/r04l_tests.log | grep -v Docs
tools/r4_ab.sh r04l 2 "c5|--config C5 --steps 20" "c5_lvl|--config C5 --steps 20 --tune 15=512" "sh3|--emulate-shard 3/8 --no-calibrate" "sh3_lvl|--emulate-shard 3/8 --no-calibrate --tune 15=512" "c3_casc|--tune 12=1" "c3|"
