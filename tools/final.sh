#!/bin/bash
# Final measurement pass of a round on one GPU box:  tools/final.sh <tag> [part]   (tag = r05, ...; part = all | tests | bench | parity | profile | misc)
# Outputs under gpurun_out/, named <tag>_*; tools/copy_profiles.sh <tag> copies the summaries into profiles/.  No step is retried.
tag=${1:?usage: tools/final.sh <tag> [part]}
part=${2:-all}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd $root
run() { name=$1; shift; timeout -k 10 600 python bench.py --no-cpu --no-emulated-shards --steps 50 --warmup 5 "$@" > $out/${tag}_bench_$name.json 2> $out/${tag}_bench_$name.err || echo "$name FAILED"; }
if [ "$part" = all ] || [ "$part" = tests ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -q > $out/${tag}_gputests.log 2>&1; tail -3 $out/${tag}_gputests.log
fi
if [ "$part" = all ] || [ "$part" = bench ]; then
  timeout -k 10 600 python bench.py --steps 200 > $out/${tag}_bench_c3.json 2> $out/${tag}_bench_c3.err; echo "c3 done"
  run c2 --config C2
  run c5 --config C5 --steps 20
  run c5_mixed --config C5 --steps 20 --precision mixed
  run c3_unfused --tune 15=1
  run c3_fused --tune 15=2
  run c3_nomerge --no-merge
  run c3_nomatrix --matrix-cores 0
  run lorentz --shape lorentz
  run doppler --shape doppler
  run phco2 --shape PHCO2 --steps 10 --warmup 2
  run shard0 --emulate-shard 0/8
  run shard3 --emulate-shard 3/8
  run shard7 --emulate-shard 7/8
  run quarter1 --emulate-shard 1/4
  run half1 --emulate-shard 1/2
  echo "benches done"
fi
if [ "$part" = all ] || [ "$part" = parity ]; then
  # BASELINE configs[4] at FULL size against the CPU port, every element (fp64 and the fp32 mixed variant)
  for k in fp64 mixed; do
    timeout -k 10 1100 python bench.py --config C5 --steps 10 --warmup 2 --cpu-stride 1 --precision $k > $out/${tag}_c5_full_$k.json 2> $out/${tag}_c5_full_$k.err || echo "c5 full $k FAILED"
  done
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
res = {}
for k in ("fp64", "mixed"):
    try:
        d = json.loads(open(f"gpurun_out/{tag}_c5_full_{k}.json").readline())
        res[k] = dict(ms_per_step=d["ms_per_step"], olr_wm2=d["olr_wm2"], dtype=d["dtype"], workload=d["config"]["workload"], cpu_baseline=d["cpu_baseline"],
                      kernel_source_sha16=d["kernel_source_sha16"])
    except Exception as e:
        res[k] = dict(error=str(e))
json.dump(res, open(f"gpurun_out/{tag}_c5_full_parity.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in (v.get("cpu_baseline") or {}).items() if kk.startswith(("max_", "olr"))} for k, v in res.items()}, indent=1))
PY
fi
if [ "$part" = all ] || [ "$part" = profile ]; then
  tools/profile.sh $tag > $out/${tag}_profile.log 2>&1; echo "profile c3 done"
  tools/profile.sh ${tag}c5 --config C5 > $out/${tag}c5_profile.log 2>&1; echo "profile c5 done"
  tools/trace_step.sh ${tag}_c3 > /dev/null 2>&1
  tools/trace_step.sh ${tag}_sh3 --emulate-shard 3/8 --no-calibrate > /dev/null 2>&1; echo "traces done"
fi
if [ "$part" = all ] || [ "$part" = misc ]; then
  python tools/mode_t_bench.py > $out/${tag}_mode_t.json 2> $out/${tag}_mode_t.err || echo "mode_t FAILED"
  python tools/multi_overlap.py $out/${tag}_multi_overlap.json > $out/${tag}_multi_overlap.log 2>&1 || echo "multi_overlap FAILED"
  python tools/shard_balance.py C3 8 C5 8 > $out/${tag}_shard_balance.json 2> $out/${tag}_shard_balance.err || echo "shard_balance FAILED"
fi
python - "$tag" <<'PY'
import json, glob, sys
tag = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/{tag}_bench_*.json")):
    try:
        d = json.loads(open(f).readline())
        print(f.split(f"{tag}_bench_")[1][:-5], "%.3f ms/step" % d["ms_per_step"], "launches", d.get("launches_per_step"), {k: round(v, 3) for k, v in d["roofline"]["kernel_ms"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
