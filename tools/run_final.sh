#!/bin/bash
# final measurements of a round on one GPU box:  tools/run_final.sh <tag>   (writes gpurun_out/<tag>_*; summaries are copied to profiles/ by hand)
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
o=$root/gpurun_out
tools/profile.sh $tag > $o/${tag}_profile.log 2>&1
python3 bench.py > $o/${tag}_bench_c3.json 2> $o/${tag}_bench_c3.err
python3 bench.py --config C2 > $o/${tag}_bench_c2.json 2>/dev/null
python3 bench.py --config C5 > $o/${tag}_bench_c5.json 2> $o/${tag}_bench_c5.err
python3 bench.py --config C5 --precision mixed --no-cpu > $o/${tag}_bench_c5_mixed.json 2>/dev/null
for sh in lorentz doppler PHCO2; do python3 bench.py --shape $sh --no-cpu --steps 5 --warmup 1 > $o/${tag}_bench_$sh.json 2>/dev/null; done
python3 bench.py --no-cpu --no-matrix-nodes > $o/${tag}_bench_c3_nomatrix.json 2>/dev/null
for r in 0 3 7; do python3 bench.py --no-cpu --emulate-shard $r/8 > $o/${tag}_bench_shard$r.json 2>/dev/null; done
python3 tools/mode_t_bench.py > $o/${tag}_mode_t.json 2>/dev/null
ls -la $o | grep $tag | wc -l
