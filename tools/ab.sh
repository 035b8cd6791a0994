#!/bin/bash
# A/B kernel builds: tools/ab.sh build/lib_a.so build/lib_b.so ...   (runs bench.py per library, prints kernel ms)
for lib in "$@"; do
  out=$(CLEARSKY_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1)
  echo "$lib $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.3f"%d["ms_per_step"], {k: round(v,3) for k,v in d["roofline"]["kernel_ms"].items()}, "OLR", d["olr_wm2"])')"
done
