timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r2u_tests.log 2>&1; tail -3 gpurun_out/r2u_tests.log
tools/ab_flags.sh "" "" 2>&1 | cut -c1-230
tools/ab_flags.sh "--config C5" "" 2>&1 | cut -c1-230
