#!/bin/bash
# A/B of kernel builds on one GPU box:  tools/ab.sh "<bench args>" lib1.so lib2.so ...   (libraries under build/, interleaved rounds)
# prints ms_per_step and the per-class kernel times of every (library, round)
args=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2 3; do
  for lib in "$@"; do
    CLEARSKY_HIP_LIB=$root/build/$lib python3 $root/bench.py --no-cpu --steps 20 --warmup 3 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernel_ms']
print('$lib round $round: %.3f ms/step  '%d['ms_per_step'] + ' '.join('%s=%.3f'%(a,b) for a,b in k.items()), ' olr=%.10f'%d['olr_wm2'])
"
  done
done
