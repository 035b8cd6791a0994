#!/bin/bash
# round 4: GPU tests + the default bench line (outputs under gpurun_out/r04b_*)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -q -x "$@" > $out/r04b_gputests.log 2>&1; tail -15 $out/r04b_gputests.log
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > $out/r04b_bench_c3.json 2> $out/r04b_bench_c3.err || { echo "bench FAILED"; tail -5 $out/r04b_bench_c3.err; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04b_bench_c3.json").readline())
print("%.3f ms/step" % d["ms_per_step"], d["roofline"]["bound"], d["roofline"]["kernel"], "tw_frac", d["roofline"]["whole_step"]["time_weighted_frac"])
print(json.dumps(d["cpu_baseline"], indent=1))
PY
