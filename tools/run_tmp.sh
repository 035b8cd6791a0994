timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r2i_tests.log 2>&1; tail -2 gpurun_out/r2i_tests.log
python bench.py > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; tail -c 3000 gpurun_out/r2i_bench.json
