#!/bin/bash
# A/B of bench.py argument sets on one GPU box, interleaved rounds:  tools/r4_ab.sh <tag> <rounds> "name1|args1" "name2|args2" ...
tag=$1; rounds=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
for round in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%|*}; args=${spec#*|}
    f=$out/${tag}_${name}_r$round.json
    timeout -k 10 300 python3 $root/bench.py --no-cpu --steps 40 --warmup 5 $args > $f 2> $out/${tag}_${name}_r$round.err || { echo "$name FAILED"; tail -3 $out/${tag}_${name}_r$round.err; continue; }
    python3 -c "
import json
d=json.loads(open('$f').readline())
k=d['roofline']['kernel_ms']
print('$name r$round: %.3f ms launches=%s  '%(d['ms_per_step'], d.get('launches_per_step')) + ' '.join('%s=%.3f'%(a,b) for a,b in k.items() if b>0.0005), ' olr=%.10f'%d['olr_wm2'])
"
  done
done
