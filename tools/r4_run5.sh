#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x > gpurun_out/r04g_tests.log 2>&1; tail -3 gpurun_out/r04g_tests.log
python tools/shard_balance.py C3 8 C5 8 > gpurun_out/r04_shard_balance.json 2> gpurun_out/r04_shard_balance.err || tail -5 gpurun_out/r04_shard_balance.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_shard_balance.json"))
for k,v in d.items():
    for p in v["passes"]:
        print(k, p["partition"][:30], "spread %.1f%%"%p["spread_pct"], "max %.3f eff %.2f"%(p["max_ms"],p["projected_efficiency"]), [round(x["ms"],3) for x in p["shards"]])
PY
