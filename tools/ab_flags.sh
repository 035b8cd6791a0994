#!/bin/bash
# A/B of bench flag sets with ONE library on one GPU box:  tools/ab_flags.sh "<common args>" "<flags A>" "<flags B>" ...   (interleaved rounds)
common=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2 3; do
  for fl in "$@"; do
    python3 $root/bench.py --no-cpu --steps 20 --warmup 3 $common $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernel_ms']
print('[$fl] round $round: %.3f ms/step  '%d['ms_per_step'] + ' '.join('%s=%.3f'%(a,b) for a,b in k.items()), ' olr=%.10f'%d['olr_wm2'])
"
  done
done
