#!/bin/bash
# Final measurement pass of round 3 on one GPU box (outputs under gpurun_out/, named r03_*; tools/copy_profiles.sh copies them to profiles/)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m pytest tests -m gpu -q > $out/r03_gputests.log 2>&1; tail -3 $out/r03_gputests.log
python bench.py --steps 200 > $out/r03_bench_c3.json 2> $out/r03_bench_c3.err; echo "c3 done"
run() { name=$1; shift; python bench.py --no-cpu --steps 50 --warmup 5 "$@" > $out/r03_bench_$name.json 2> $out/r03_bench_$name.err || echo "$name FAILED"; }
run c2 --config C2
run c5 --config C5 --steps 20
run c5_mixed --config C5 --steps 20 --precision mixed
run c3_nomerge --no-merge
run c3_nomatrix --matrix-cores 0
run lorentz --shape lorentz
run doppler --shape doppler
run phco2 --shape PHCO2 --steps 10 --warmup 2
run phco2_pointwise --shape PHCO2 --steps 3 --warmup 1 --no-interp
run shard0 --emulate-shard 0/8
run shard3 --emulate-shard 3/8
run shard7 --emulate-shard 7/8
run quarter1 --emulate-shard 1/4
run half1 --emulate-shard 1/2
echo "benches done"
tools/profile.sh r03 > $out/r03_profile.log 2>&1; echo "profile done"
tools/run_ubench.sh > $out/r03_ubench.log 2>&1; echo "ubench done"
python tools/mode_t_bench.py > $out/r03_mode_t.json 2> $out/r03_mode_t.err || echo "mode_t FAILED"
python tools/multi_overlap.py > $out/r03_multi_overlap.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_bench_*.json")):
    try:
        d = json.loads(open(f).readline())
        print(f.split("r03_bench_")[1][:-5], "%.3f ms/step" % d["ms_per_step"], "launches", d.get("launches_per_step"), {k: round(v, 3) for k, v in d["roofline"]["kernel_ms"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
