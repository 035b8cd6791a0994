#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_flux_fused.py tests/test_gpu_interp.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r04j_tests.log 2>&1; tail -3 gpurun_out/r04j_tests.log | grep -v Docs
python tools/flux_phases.py shard; python tools/flux_phases.py C2
tools/r4_ab.sh r04j 2 "sh3|--emulate-shard 3/8 --no-calibrate" "sh3_nc|--emulate-shard 3/8 --no-calibrate --tune 15=256" "sh0|--emulate-shard 0/8 --no-calibrate" "sh7|--emulate-shard 7/8 --no-calibrate" "q1|--emulate-shard 1/4 --no-calibrate" "q1_nc|--emulate-shard 1/4 --no-calibrate --tune 15=256" "c2|--config C2"
