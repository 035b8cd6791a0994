#!/bin/bash
# interleaved A/B of bench.py variants in ONE gpurun call:  tools/ab.sh <tag> <rounds> "<common bench args>" "<variant args>" ["<variant args>" ...]
# -> gpurun_out/<tag>_v<i>_r<j>.json and a table of ms/step + per-class kernel times on stdout
tag=$1; rounds=$2; common=$3; shift 3
root=${GRAFT_REPO_ROOT:-$PWD}
i=0
for v in "$@"; do vars[$i]="$v"; i=$((i+1)); done
for r in $(seq 1 $rounds); do
  for i in "${!vars[@]}"; do
    timeout -k 10 300 python3 $root/bench.py --no-cpu $common ${vars[$i]} > $root/gpurun_out/${tag}_v${i}_r${r}.json 2> $root/gpurun_out/${tag}_v${i}_r${r}.err || { echo "variant $i failed"; tail -5 $root/gpurun_out/${tag}_v${i}_r${r}.err; }
  done
done
python3 - "$root/gpurun_out/$tag" "$rounds" "${vars[@]}" <<'PY'
import json, sys
base, rounds, vs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for i, v in enumerate(vs):
    ms, km = [], None
    for r in range(1, rounds + 1):
        try:
            d = json.load(open(f"{base}_v{i}_r{r}.json"))
            ms.append(d["ms_per_step"]); km = d["roofline"]["kernel_ms"]; olr = d["olr_wm2"]
        except Exception as e:
            ms.append(float("nan"))
    print(f"v{i} [{v}] ms/step " + " ".join(f"{m:.4f}" for m in ms) + (f" | olr {olr:.10f} | " + " ".join(f"{k}={x*1e3:.0f}" for k, x in km.items()) if km else ""))
PY
