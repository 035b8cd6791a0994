tools/ab.sh "" lib_far6.so lib_nzfix.so 2>&1 | cut -c1-200
tools/ab.sh "--config C5" lib_far6.so lib_nzfix.so 2>&1 | cut -c1-200
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r2l_tests.log 2>&1; tail -2 gpurun_out/r2l_tests.log
