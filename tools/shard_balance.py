#!/usr/bin/env python3
"""Emulated shard times of a configuration on ONE GPU: tools/shard_balance.py C3 8 [C5 8 ...]  -> JSON on stdout.
Every range of an N-way partition is timed as bench.py times a shard (its own process, --nu-range a:b): what rank r of an N-GPU run
would take before the all-reduce.  First the cost model's partition (cs_balanced_ranges), then the partition re-cut from those measured
times (cs_rebalance_ranges -- what bench.py does across ranks before its timed region, and cs_fluxes_discretized_multi inside its first
call), then once more from the second set of times."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)


def bench(cfg, extra):
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--no-cpu", "--steps", "30", "--warmup", "5"] + extra,
                       capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode or not line:
        raise RuntimeError(p.stderr[-400:])
    return json.loads(line[0])


def time_partition(cfg, ranges):
    rows = []
    for a, b in ranges:
        d = bench(cfg, ["--nu-range", f"{a}:{b}"])
        rows.append(dict(range=[a, b], ms=d["ms_per_step"], launches=d["launches_per_step"],
                         kernel_ms={k: round(v, 4) for k, v in d["roofline"]["kernel_ms"].items()}))
    ms = [x["ms"] for x in rows]
    return dict(shards=rows, min_ms=min(ms), max_ms=max(ms), mean_ms=sum(ms) / len(ms), spread_pct=100.0 * (max(ms) - min(ms)) / (sum(ms) / len(ms)))


def main():
    import clearsky_jl_amd as cs
    import workloads as W
    out = {}
    args = sys.argv[1:]
    for cfg, n in zip(args[0::2], args[1::2]):
        n = int(n)
        c = W.config(cfg)
        lines_pos = [a.sl.nu for a in c["absorbers"] if isinstance(a, cs.DirectGas)]
        full_ms = bench(cfg, [])["ms_per_step"]
        passes = []
        ranges = W.balanced_ranges(c["nu"], c["absorbers"], n)
        for it in range(3):
            res = time_partition(cfg, ranges)
            res["partition"] = "cost model (cs_balanced_ranges)" if it == 0 else f"re-cut from measured times, pass {it} (cs_rebalance_ranges)"
            res["projected_efficiency"] = full_ms / n / res["max_ms"]
            passes.append(res)
            times = [x["ms"] for x in res["shards"]]
            ranges = cs.rebalance_ranges(c["nu"], lines_pos, ranges, times, fixed_time=0.3 * min(times))
        out[f"{cfg}/{n}"] = dict(full_grid_ms=full_ms, passes=passes,
                                 note="projected_efficiency = (full-grid step / N) / slowest shard: an N-GPU run before its all-reduce, EMULATED on one "
                                      "GPU, one shard at a time")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
