timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_interp.py -m gpu -x -q > gpurun_out/r2s_tests.log 2>&1; tail -3 gpurun_out/r2s_tests.log
tools/ab_flags.sh "" "" 2>&1 | cut -c1-230
tools/ab_flags.sh "--config C5" "" 2>&1 | cut -c1-230
tools/ab_flags.sh "--emulate-shard 0/8" "" 2>&1 | cut -c1-230
