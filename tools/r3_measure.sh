#!/bin/bash
# round-3 measurement pass on one GPU box: tools/r3_measure.sh <tag> [extra bench args]
# writes gpurun_out/<tag>_*.json|log ; one line per configuration on stdout
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
show() { python3 -c "
import json,sys
d=json.loads(open('$1').readline())
k=d['roofline']['kernel_ms']
print('$2: %.3f ms/step launches=%s groups=%s  '%(d['ms_per_step'], d.get('launches_per_step'), d.get('launch_groups')) + ' '.join('%s=%.3f'%(a,b) for a,b in k.items()), ' olr=%.10f'%d['olr_wm2'])
"; }
run() { name=$1; shift; python3 $root/bench.py --no-cpu --steps 50 --warmup 5 "$@" > $out/${tag}_$name.json 2> $out/${tag}_$name.err && show $out/${tag}_$name.json $name; }
run c3 "$@" &&
run c3_nomerge --no-merge "$@" &&
run c5 --config C5 --steps 20 "$@" &&
run c5_nomerge --config C5 --steps 20 --no-merge "$@" &&
run shard0 --emulate-shard 0/8 "$@" &&
run shard3 --emulate-shard 3/8 "$@" &&
run shard7 --emulate-shard 7/8 "$@" &&
run shard7_nomerge --emulate-shard 7/8 --no-merge "$@" &&
run c2 --config C2 "$@"
