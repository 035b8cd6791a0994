#!/bin/bash
# Baseline pass at the start of round 4: GPU tests + the bench lines the round works on (outputs under gpurun_out/r04a_*)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/r04a_gputests.log 2>&1; tail -3 $out/r04a_gputests.log
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu --steps 50 --warmup 5 "$@" > $out/r04a_bench_$name.json 2> $out/r04a_bench_$name.err || echo "$name FAILED"; }
run c3
run c2 --config C2
run c5 --config C5 --steps 20
run shard0 --emulate-shard 0/8
run shard3 --emulate-shard 3/8
run shard7 --emulate-shard 7/8
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04a_bench_*.json")):
    try:
        d = json.loads(open(f).readline())
        print(f.split("r04a_bench_")[1][:-5], "%.3f ms/step" % d["ms_per_step"], "launches", d.get("launches_per_step"), {k: round(v, 3) for k, v in d["roofline"]["kernel_ms"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
