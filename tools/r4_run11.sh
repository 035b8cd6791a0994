#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
for r in 0 1 2 3 4 5 6 7; do
  python bench.py --no-cpu --steps 50 --warmup 5 --emulate-shard $r/8 > gpurun_out/r04m_cal_$r.json 2>/dev/null
  python bench.py --no-cpu --steps 50 --warmup 5 --emulate-shard $r/8 --no-calibrate > gpurun_out/r04m_nocal_$r.json 2>/dev/null
done
python - <<'PY'
import json
for tag in ("cal", "nocal"):
    ms = [json.loads(open(f"gpurun_out/r04m_{tag}_{r}.json").readline())["ms_per_step"] for r in range(8)]
    print(tag, [round(x, 3) for x in ms], "max %.3f spread %.1f%%" % (max(ms), 100 * (max(ms) - min(ms)) / (sum(ms) / 8)))
d = json.loads(open("gpurun_out/r04m_cal_3.json").readline())["partition"]
print(d["model_shard_ms"]); print(d["model_ranges"]); print(d["ranges"])
PY
