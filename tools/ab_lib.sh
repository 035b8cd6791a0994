#!/bin/bash
# interleaved A/B of TWO BUILDS of the library in one gpurun call (variants a compile-time switch separates):
#   tools/ab_lib.sh <tag> <rounds> <other.so> "<bench args>" ["<bench args>" ...]
# the in-tree library is build A; <other.so> (built beforehand, e.g. under build/) is swapped in for B and out again.  Scratch copy only.
tag=$1; rounds=$2; other=$3; shift 3
root=${GRAFT_REPO_ROOT:-$PWD}
lib=$root/clearsky.jl_amd/csrc/libclearsky_hip.so
mkdir -p $root/build; cp $lib $root/build/_lib_A.so
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then cp $root/build/_lib_A.so $lib; else cp $other $lib; fi
    i=0
    for args in "$@"; do
      timeout -k 10 300 python3 $root/bench.py --no-cpu --no-emulated-shards --steps 50 --warmup 5 $args > $root/gpurun_out/${tag}_${v}${i}_r${r}.json 2> $root/gpurun_out/${tag}_${v}${i}_r${r}.err || echo "$v $args failed"
      i=$((i+1))
    done
  done
done
cp $root/build/_lib_A.so $lib
python3 - "$root/gpurun_out/$tag" "$rounds" "$@" <<'PY'
import json, sys
base, rounds, argsets = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for i, a in enumerate(argsets):
    for v in "AB":
        ms, km = [], None
        for r in range(1, rounds + 1):
            try:
                d = json.load(open(f"{base}_{v}{i}_r{r}.json")); ms.append(d["ms_per_step"]); km = d["roofline"]["kernel_ms"]; olr = d["olr_wm2"]
            except Exception:
                ms.append(float("nan"))
        print(f"{v} [{a}] ms/step " + " ".join(f"{m:.4f}" for m in ms) + (f" | olr {olr:.10f} | " + " ".join(f"{k}={x*1e3:.0f}" for k, x in km.items()) if km else ""))
PY
