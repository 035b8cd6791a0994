timeout -k 10 400 python -m pytest tests/test_gpu_interp.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r2k_tests.log 2>&1; tail -2 gpurun_out/r2k_tests.log
tools/ab_flags.sh "" "" 2>&1 | cut -c1-190
tools/ab_flags.sh "--config C5" "" 2>&1 | cut -c1-190
