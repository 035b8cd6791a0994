timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r2h_tests.log 2>&1; tail -2 gpurun_out/r2h_tests.log
tools/ab_flags.sh "" "" 2>&1 | cut -c1-150
tools/ab_flags.sh "--config C5" "" "--no-matrix-nodes" 2>&1 | cut -c1-150
tools/ab_flags.sh "--emulate-shard 0/8" "" 2>&1 | cut -c1-150
