#!/bin/bash
# tools/ab_shard.sh R/N lib...  : kernel times of an emulated shard per library
sh=$1; shift
for lib in "$@"; do
  CLEARSKY_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --emulate-shard $sh --steps 30 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib shard $sh ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()})"
done
