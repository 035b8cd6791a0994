#!/usr/bin/env python3
"""bench.py -- spectral-points/s of the line-by-line hot path on N MI355X (BASELINE.json metric).

A "step" is one whole-column evaluation (line sums at every node -> tau -> Planck/multi-stream fluxes -> band
integrals) with all inputs already resident in HBM.  Workload at N=1: BASELINE.json configs[2] -- H2O+CO2 Earth-like
column, 1e5 wavenumbers x 60 layers, Voigt, fp64, on the ~1e5-line seeded synthetic table (the
container only holds 8 657 real H2O+CO2 lines).  For N>1 the SAME column is sharded over contiguous, work-balanced
wavenumber ranges (strong scaling, configs[3]) and the band fluxes are combined by one RCCL all-reduce of 2*np doubles.

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

# VALU instructions per (lane, line) of each loop body, counted in the gfx950 ISA of k_voigt_far / k_cheb_nodes (hipcc -S, loops
# unrolled by 4: 52/68/64/80/96/148 per four lines; profiles/r02_notes.md).  The 2-/3-/4-term far-wing bodies are the same
# instructions in both kernels; "_cut" adds the cut-off compare + select (and |dnu|).
VALU_PER_LINE = dict(t2=13, t2_cut=17, t3=16, t3_cut=20, t4=20, t4_cut=24, near_zone=37)
# k_voigt_near, per (nu, line, state) pair: the evaluation block of the ISA (basic block of the trip loop: 64 VALU in tier 0, 407 in
# tier 1) + 41 for the queue write, the record fetch of the next trip and the ordered per-lane sum (profiles/r03_notes.md)
VALU_NEAR = dict(tier0_eval=64, tier1_eval=407, per_candidate=41)
# the flux sweeps per (wavenumber, layer), ISA of k_rt<5, true> (tools/isa_loops.py): 334 VALU in the downward loop (optical depth, Planck, five
# exp + layerplanck, wave reduction), 286 in the upward one; other stream counts: 60 + 55 per stream and sweep
VALU_RT = dict(down=334, up=286, per_stream=55, fixed=60)
VALU_ISSUE_PEAK = 256 * 4 * 16 * 2.4e9     # fp64-rate lane-instructions per second: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: the fp64 matrix path has the vector unit's peak on this part


def source_stamp():
    """Build id of the LOADED library (cs_build_id: the sha256 of the sources it was compiled from, baked in at build time): PMC
    traffic collected offline is only quoted while profiles/pmc_traffic.json carries the id of this very binary."""
    import clearsky_jl_amd as cs
    return cs.lib().cs_build_id().decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # (0.55 s of timed region at C3: long enough for an outside utilisation sampler to see it)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--lines", default=None, help="synthetic | fixture")
    ap.add_argument("--nnu", type=int, default=None)
    ap.add_argument("--shape", default="voigt", help="line shape of every gas: voigt (headline) | lorentz | doppler | PHCO2")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--precision", default="fp64", help="fp64 (headline) | mixed (fp32 far wings, BASELINE configs[4])")
    ap.add_argument("--no-interp", action="store_true", help="evaluate every (nu, line) pair (no far-wing interpolation)")
    ap.add_argument("--interp-first-level", type=int, default=-1, help="tuning: first interval level every gas uses (-1 = by line density)")
    ap.add_argument("--no-matrix-nodes", action="store_true", help="tuning: no far-line sums on the matrix cores (same as --matrix-cores 0)")
    ap.add_argument("--matrix-cores", type=int, default=1, help="tuning: far-line sums on the matrix cores: 1 where the grid is long enough (default), 2 always, 0 never")
    ap.add_argument("--no-merge", action="store_true", help="tuning: one launch set per gas instead of one merged line table per column")
    ap.add_argument("--tune", default="", help="tuning switches k=v[,k=v...] (cs_set_tuning)")
    ap.add_argument("--far-s", type=float, default=1e6, help="mixed precision: x^2 threshold of the fp32 region")
    ap.add_argument("--emulate-shard", default=None, help="R/N: time only shard R of an N-way split on this one GPU (rehearsal)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL, default) | gloo (rehearsal of N>1 on fewer GPUs)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and all-reduce the band fluxes even with one rank "
                    "(rehearsal of the RCCL path on a one-GPU box)")
    ap.add_argument("--nu-range", default=None, help="a:b -- time only the wavenumbers [a, b) of the grid, with the global trapezoid weights (rehearsal of one shard)")
    ap.add_argument("--no-calibrate", action="store_true", help="N > 1: keep the cost model's partition instead of re-cutting it from the shards' measured times")
    ap.add_argument("--no-emulated-shards", action="store_true", help="skip the emulated_shards block (shards 0, 3, 7 of 8 and 1 of 4 timed after the headline)")
    ap.add_argument("--cpu-stride", type=int, default=0, help="cpu_baseline evaluates every n-th wavenumber (0 = size the sample for ~15 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as fresh child processes, BEFORE anything
        # in this process touches the GPU (no exec of a GPU-initialised process), and exit with the launcher's code
        # (--standalone: the launcher's own c10d store picks a free port and keeps it -- no probe-close-reuse of a port number)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"error: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    N = world

    import torch
    import clearsky_jl_amd as cs
    import workloads as W

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the HIP path)")
    ndev = torch.cuda.device_count()
    dev = local_rank % ndev          # gloo rehearsals may put several ranks on one card
    torch.cuda.set_device(dev)
    dist = None
    use_dist = N > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        init = {}
        if "RANK" not in os.environ:     # --force-dist without a launcher: a one-rank group rendezvousing through a file (no port to pick)
            import tempfile
            init = dict(init_method="file://" + os.path.join(tempfile.mkdtemp(prefix="cs_bench_"), "rdzv"), rank=0, world_size=1)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), **init)
        else:
            dist.init_process_group(args.dist_backend, **init)

    cfg = W.config(args.config, nnu=args.nnu, lines_kind=args.lines, shape=args.shape)
    nu, nl = cfg["nu"], cfg["nl"]
    ranges = W.balanced_ranges(nu, cfg["absorbers"], N)
    if args.emulate_shard:
        r_, n_ = map(int, args.emulate_shard.split("/"))
        ranges = [W.balanced_ranges(nu, cfg["absorbers"], n_)[r_]]
    if args.nu_range:
        ranges = [tuple(int(x) for x in args.nu_range.split(":"))]
    ctx = cs.Context(dev)
    ctx.set_precision(args.precision, args.far_s)
    ctx.set_interp(not args.no_interp)
    ctx.set_interp_plan(first_level=args.interp_first_level)
    ctx.set_matrix_cores(0 if args.no_matrix_nodes else args.matrix_cores)
    ctx.set_merge(not args.no_merge)
    for kv in filter(None, args.tune.split(",")):
        ctx.set_tuning(*map(int, kv.split("=")))
    def make_column(rng):
        return cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                         theta_s=cfg["theta_s"], want_tau=True, want_M=True, nu_range=rng, ctx=ctx)

    def shard_ms(c, reps=12, batches=4):
        """what a shard takes per step: warm-up, then the fastest of a few batches (clock ramps and a cold first batch only ever add)"""
        for _ in range(6):
            c.run()
        c.sync()
        best = float("inf")
        for _ in range(batches):
            t1 = time.perf_counter()
            for _ in range(reps):
                c.run()
            c.sync()
            best = min(best, (time.perf_counter() - t1) * 1e3 / reps)
        return best

    t_setup = time.perf_counter()
    col = make_column(ranges[rank])
    col.sync()
    setup_ms = (time.perf_counter() - t_setup) * 1e3   # one-off: closures, windows, interpolation matrices, workspaces, uploads
    # N > 1: the partition came from a cost model whose constants were fitted to BASELINE configs[2]; before the timed region it is re-cut
    # ONCE from what the shards are measured to take on THIS column (cs_rebalance_ranges; one all-reduce of N doubles at setup time -- the
    # path's only per-step collective stays the band-flux all-reduce).  --emulate-shard r/N measures all N shards on the one GPU for that.
    partition = None
    n_parts = N if not args.emulate_shard else int(args.emulate_shard.split("/")[1])
    lines_pos = [a.sl.nu for a in cfg["absorbers"] if isinstance(a, cs.DirectGas)]
    if n_parts > 1 and not args.no_calibrate and args.emulate_shard:
        try:
            base = W.balanced_ranges(nu, cfg["absorbers"], n_parts)
            # every shard through the SAME context, one after the other (a second context's streams would share hardware queues with
            # this one's and lose their overlap: measured +12 %), the resident column being replaced each time
            me = int(args.emulate_shard.split("/")[0])
            times = []
            for r_i in range(n_parts):
                tmp = make_column(base[r_i])
                times.append(shard_ms(tmp))
                del tmp
            recut = cs.rebalance_ranges(nu, lines_pos, base, times, fixed_time=0.3 * min(times))
            partition = dict(model_ranges=base, model_shard_ms=times, ranges=recut, calibrated=recut != base)
            col = make_column(recut[me])
            col.sync()
        except Exception as exc:     # (one process: a failed calibration keeps the model's partition, the step itself does not depend on it)
            partition = dict(calibrated=False, error=repr(exc))
            col = make_column(ranges[0])
            col.sync()
    elif n_parts > 1 and not args.no_calibrate:
        # N ranks.  Every decision here is COLLECTIVE: each rank reaches the same all-reduces in the same order whatever happened to it
        # locally, and the re-cut partition is adopted only if EVERY rank measured its shard and set its new one up -- otherwise every rank
        # keeps the model's range.  (A rank falling back on its own would leave ranges that do not tile the grid -- wavenumbers counted
        # twice or not at all under a healthy-looking JSON line -- or pair its band-flux all-reduce with the others' calibration one.)
        cdev = f"cuda:{dev}" if args.dist_backend == "nccl" else "cpu"
        base = ranges
        tv = torch.zeros(n_parts + 1, dtype=torch.float64, device=cdev)     # [times of the N shards..., number of ranks that failed]
        err = None
        try:
            if os.environ.get("CS_BENCH_FAIL_CALIBRATION") == f"measure:{rank}":
                raise RuntimeError("injected calibration failure (test)")
            tv[rank] = shard_ms(col)
        except Exception as exc:
            err = repr(exc)
            tv[n_parts] = 1.0
        dist.all_reduce(tv)
        times = [float(x) for x in tv[:n_parts].cpu()]
        adopted = False
        if float(tv[n_parts].item()) == 0.0:
            recut = cs.rebalance_ranges(nu, lines_pos, base, times, fixed_time=0.3 * min(times))      # (same inputs on every rank: same cut)
            ok = torch.ones(1, dtype=torch.float64, device=cdev)
            new_col = None
            try:
                if os.environ.get("CS_BENCH_FAIL_CALIBRATION") == f"setup:{rank}":
                    raise RuntimeError("injected set-up failure (test)")
                if recut[rank] != base[rank]:
                    new_col = make_column(recut[rank])     # (replaces this context's resident column)
                    new_col.sync()
            except Exception as exc:
                err = repr(exc)
                ok[0] = 0.0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 1.0:
                adopted = True
                ranges = recut
                if new_col is not None:
                    col = new_col
            else:       # somebody could not set its new shard up: everybody back to the model's range
                del new_col
                col = make_column(base[rank])
                col.sync()
        partition = dict(model_ranges=base, model_shard_ms=times, ranges=ranges, calibrated=adopted and ranges != base)
        if not adopted:
            partition["error"] = err or "another rank failed to calibrate: every rank keeps the model's partition"
    if use_dist and N > 1:
        # what each rank ACTUALLY runs, gathered: the ranges must tile [0, nnu) exactly once
        mine = torch.tensor([col.j0, col.j1], dtype=torch.int64, device=(f"cuda:{dev}" if args.dist_backend == "nccl" else "cpu"))
        allr = [torch.zeros_like(mine) for _ in range(N)]
        dist.all_gather(allr, mine)
        by_rank = [[int(x[0]), int(x[1])] for x in allr]
        tiles = by_rank[0][0] == 0 and by_rank[-1][1] == len(nu) and all(by_rank[i][1] == by_rank[i + 1][0] for i in range(N - 1))
        if partition is None:
            partition = dict(calibrated=False)
        partition["ranges_by_rank"] = by_rank
        if not tiles:
            raise SystemExit(f"rank {rank}: the ranks' wavenumber ranges {by_rank} do not tile [0, {len(nu)})")
    # one explicit torch stream carries the kernels, the D2D copy of the band fluxes and the collective, so they are
    # ordered by the stream (torch's default stream has handle 0, which the C ABI reads as "use the context's stream")
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    F = torch.zeros(2 * col.np, dtype=torch.float64, device=f"cuda:{dev}")

    col.set_flux_dst(F.data_ptr())      # the flux kernel's last blocks write the band fluxes where the collective reduces them in place

    def step():
        col.run(stream)
        if use_dist:
            if args.dist_backend == "nccl":
                dist.all_reduce(F)   # RCCL over xGMI: 2*np doubles, the only collective of the path
            else:
                Fc = F.cpu()
                dist.all_reduce(Fc)
                F.copy_(Fc)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    points = len(nu) * nl
    value = points * args.steps / dt
    Fh = F.cpu().numpy()
    olr = float(Fh[0])

    # per-kernel HIP-event timing on the launch stream (rank-local) and the roofline of every line-stage kernel class
    # (cs_column_profile runs the step on ONE stream with events between the classes; the timed loop above overlaps the node
    # kernels with the per-point ones on a side stream, so the classes add up to a little more than ms_per_step)
    prof = col.profile(reps=max(3, min(10, args.steps)), stream=stream)
    cnt = col.counts()
    info = col.info()
    ngrp = max(int(info["groups"]), 1)     # launch sets per step: the gases of a column that share a cut-off run as ONE merged table
    K = col.K
    lines_total = sum(len(g.sl.nu) for g in col.gases)
    work = col.work()
    interp_on = work["levels"] > 0
    nt64 = (col.nnu + 63) // 64
    sec = lambda name: prof.get(name, 0.0) * 1e-3 / ngrp      # seconds per launch of a class

    def mfma_entry(name, useful, issued, note, alg_bytes=None, requested=None):
        t = sec(name)
        d = dict(bound="mfma", ms_per_launch=t * 1e3, launches_per_step=ngrp, useful_flops_per_launch=useful / ngrp,
                 issued_flops_per_launch=issued / ngrp, achieved=(useful / ngrp / t / 1e12) if t > 0 else 0.0,
                 achieved_issued=(issued / ngrp / t / 1e12) if t > 0 else 0.0, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                 frac=(useful / ngrp / t / 1e12 / FP64_MFMA_PEAK_TFLOPS) if t > 0 else 0.0,
                 frac_issued=(issued / ngrp / t / 1e12 / FP64_MFMA_PEAK_TFLOPS) if t > 0 else 0.0, note=note)
        if alg_bytes is not None:
            # what must cross HBM once: every per-(state, line) record in range (32 B) + the read-modify-write of the plane the kernel adds to;
            # requested: the records of every piece as the waves ask for them (neighbouring tiles / intervals share lines: L2 traffic)
            d.update(algorithmic_bytes_per_launch=alg_bytes / ngrp, requested_record_bytes_per_launch=(requested or 0) / ngrp,
                     hbm_frac_of_algorithmic_bytes=(alg_bytes / ngrp / t / 1e9 / HBM_PEAK_GBS) if t > 0 else 0.0)
        return d

    def valu_entry(name, lane_instr, extra):
        t = sec(name)
        d = dict(bound="valu_issue", ms_per_launch=t * 1e3, launches_per_step=ngrp, lane_instructions_per_launch=lane_instr / ngrp,
                 achieved=(lane_instr / ngrp / t) if t > 0 else 0.0, peak=VALU_ISSUE_PEAK, unit="fp64-rate lane-instructions/s",
                 frac=(lane_instr / ngrp / t / VALU_ISSUE_PEAK) if t > 0 else None)
        d.update(extra)
        return d

    kern = {}
    # matrix-core kernels: USEFUL flops = 2 x series terms per (point | node, line, state) with the point inside the cut-off and outside
    # the core radius and the state a real one (cs_column_work counts them on the host from the zone tables and the grid); ISSUED = 2048
    # per matrix instruction, i.e. masked columns, padded states and the fill of the last 4-line step included
    rec_unique = 32.0 * K * cnt["lines_in_range"]          # every per-(state, line) record some window of this grid can reach, once
    kern["k_voigt_edge_mx"] = mfma_entry("far_mx", work["edge_mx_flops_useful"], work["edge_mx_flops_issued"],
                                         "window ends, pieces between interpolated sets and near zone, window cores beyond the series radius",
                                         alg_bytes=rec_unique + 16.0 * K * col.nnu, requested=work.get("edge_mx_record_bytes_requested"))
    kern["k_cheb_nodes_mx"] = mfma_entry("nodes_mx", work["nodes_mx_flops_useful"], work["nodes_mx_flops_issued"],
                                         "node sums of the interpolated far wings, 3- / 4-term series in 1/dnu^2",
                                         alg_bytes=rec_unique + 16.0 * 64 * max(work["intervals"], 0) * (16 * ((K + 15) // 16)),
                                         requested=work.get("nodes_mx_record_bytes_requested"))
    if prof.get("apply", 0.0) > 0 and not col.baked and not col.U.cia:
        t = prof["apply"] * 1e-3
        kern["k_cheb_apply_mfma"] = dict(bound="mfma", ms_per_launch=t * 1e3, launches_per_step=1, useful_flops_per_launch=work["apply_flops"],
                                         achieved=work["apply_flops"] / t / 1e12 if t > 0 else 0.0, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                         frac=work["apply_flops"] / t / 1e12 / FP64_MFMA_PEAK_TFLOPS if t > 0 else 0.0,
                                         note="node sums -> grid: C[level](64 nodes x points) . F(64 nodes x states), once per step for all gases")
    # vector kernels: lane-instructions = sum over loop bodies of (lines x 64 lanes) x VALU instructions per line, read off the ISA
    # (hipcc -S, gfx950; all 64 lanes of a wave count, also those the cut-off predicate masks), against the fp64-rate issue peak
    db, nb = work.get("direct_by_body", {}), work.get("node_by_body", {})
    kern["k_voigt_far"] = valu_entry("far", sum(db.get(b_, 0) * VALU_PER_LINE[b_] for b_ in db), dict(lines_x_lanes_by_body=db))
    line_kernel = {0: "k_voigt_far", 1: f"k_linesum<{args.shape}>", 2: "k_phco2"}[int(info.get("line_kernel", 0))]
    kern["k_voigt_far"]["kernel"] = line_kernel       # (the class "far" is whatever summed the per-point far lines: Doppler and PHCO2 have their own kernels)
    if line_kernel != "k_voigt_far":
        kern["k_voigt_far"].update(frac=None, lane_instructions_per_launch=None, achieved=None,
                                   note="the instruction accounting above is k_voigt_far's: not applicable to this kernel, only its time is reported")
    kern["k_cheb_nodes"] = valu_entry("nodes", sum(nb.get(b_, 0) * VALU_PER_LINE[b_] for b_ in nb), dict(lines_x_nodes_by_body=nb))
    kern["k_voigt_sub"] = valu_entry("sub", work.get("sub_evals", 0) * VALU_PER_LINE["near_zone"], dict(lane_line_evals=work.get("sub_evals", 0)))
    p0, p1 = work.get("near_pairs_tier0", 0), work.get("near_pairs_tier1", 0)
    kern["k_voigt_near"] = valu_entry("near", p0 * (VALU_NEAR["tier0_eval"] + VALU_NEAR["per_candidate"]) + p1 * (VALU_NEAR["tier1_eval"] + VALU_NEAR["per_candidate"]),
                                      dict(pairs_tier0=p0, pairs_tier1=p1, valu_per_pair=VALU_NEAR,
                                           launches_per_step=int(info.get("near_launches", ngrp)),     # (counted by the library where it launches them)
                                           note="tier 0 (100 <= x^2+y^2 < 1e3: continued fraction) + tier 1 (< 100: trapezoid + pole): k_voigt_near_both, one "
                                                "launch, or k_voigt_near<0> + <1> on grids of half a million (tile, state) waves and more; (nu, line, state) "
                                                "pairs counted on the host from the device's per-(state, line) records; both tiers in one time"))
    # HBM-bound kernels: algorithmic bytes = what must cross HBM once
    setup_bytes = (58.0 * lines_total + 48.0 * K * lines_total + 32.0 * K * nt64 + 48.0 * K * work["intervals"]) / ngrp
    t = sec("prep")
    kern["k_gas_setup"] = dict(bound="hbm", ms_per_launch=t * 1e3, launches_per_step=ngrp, algorithmic_bytes_per_launch=setup_bytes,
                               achieved=setup_bytes / t / 1e9 if t > 0 else 0.0, peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=setup_bytes / t / 1e9 / HBM_PEAK_GBS if t > 0 else 0.0,
                               note="58 B per line read + 48 B per (state, line) record written + zone tables; the time includes k_mxzones")
    rt_bytes = 8.0 * col.nnu * (K + (col.nl if col.want_tau else 0) + (2 * col.np if col.want_M else 0) + 2)
    t = prof["rt"] * 1e-3
    ns_ = cfg["core"].nstream
    rt_valu = (VALU_RT["down"] + VALU_RT["up"]) if ns_ == 5 else 2 * (VALU_RT["fixed"] + VALU_RT["per_stream"] * ns_)
    rt_lane_instr = float(col.nnu) * col.nl * rt_valu
    flux_name = {0: "k_rt", 2: "k_flux_chunk", 3: "k_flux_scan"}[int(info.get("flux_form", 0))]
    kern["k_rt"] = dict(bound="valu_issue", kernel=flux_name, ms_per_launch=t * 1e3, launches_per_step=1, lane_instructions_per_launch=rt_lane_instr,
                        valu_per_point_layer=rt_valu, achieved=rt_lane_instr / t if t > 0 else 0.0, peak=VALU_ISSUE_PEAK,
                        unit="fp64-rate lane-instructions/s", frac=rt_lane_instr / t / VALU_ISSUE_PEAK if t > 0 else 0.0,
                        algorithmic_bytes_per_launch=rt_bytes, hbm_achieved_gbs=rt_bytes / t / 1e9 if t > 0 else 0.0,
                        hbm_frac=rt_bytes / t / 1e9 / HBM_PEAK_GBS if t > 0 else 0.0,
                        note="the flux sweeps: 2 ns + 1 exponentials (own 20-instruction exp), layerplanck and a wave reduction per (nu, layer) and sweep -- "
                             "priced against the fp64 issue peak from the ISA's loop bodies; the HBM figure beside it (cross-sections read, tau, M+, M- "
                             "written when asked for).  k_flux_* also finish the cross-sections on chip (interpolated wings on the matrix cores, CIA "
                             "pairs, near-line plane): that work is overhead in this fraction")
    # the dominant kernel = the class with the largest time per step
    cls = dict(k_voigt_edge_mx="far_mx", k_cheb_nodes_mx="nodes_mx", k_voigt_far="far", k_cheb_nodes="nodes", k_voigt_sub="sub",
               k_voigt_near="near", k_gas_setup="prep", k_rt="rt")
    dom = max(cls, key=lambda kname: prof.get(cls[kname], 0.0))
    D = kern[dom]
    # PMC traffic cannot be collected inside this process (rocprofv3 --pmc wraps the program, FETCH_SIZE and WRITE_SIZE in
    # separate passes: tools/profile.sh writes profiles/pmc_traffic.json).  It is quoted only when that file was measured on the
    # library loaded now (same build id) and on this exact workload; otherwise null -- never a stale number.
    traffic, traffic_note, traffic_step = None, "no PMC profile of this build/workload under profiles/", None
    try:
        pm = json.load(open(os.path.join(_ROOT, "profiles", "pmc_traffic.json" if args.config == "C3" else f"pmc_traffic_{args.config.lower()}.json")))
        default_wl = (args.nnu is None and args.lines is None and args.shape == "voigt" and N == 1 and interp_on and args.precision == "fp64"
                      and not args.emulate_shard and not args.nu_range and args.tune in ("", "2=0", "2=0,7=0") and not args.no_merge)
        if pm.get("source_sha16") != source_stamp():
            traffic_note = f"profiles/pmc_traffic.json belongs to build {pm.get('source_sha16')}, the loaded library is {source_stamp()}"
        elif pm.get("config") != args.config or not default_wl:
            traffic_note = "profiles/pmc_traffic.json was measured on another workload"
        else:
            dom_name = D.get("kernel", dom)   # (the flux class runs as k_rt, k_flux_scan or k_flux_chunk[3] depending on the grid)
            match = [k_ for k_ in pm["kernels"] if k_.split("<")[0] == dom_name or (dom_name == "k_flux_chunk" and k_.startswith("k_flux_chunk"))]
            if match:
                kk = pm["kernels"][match[0]]
                traffic = (kk["FETCH_SIZE_KB"] + kk["WRITE_SIZE_KB"]) * 1024.0
                traffic_note = pm.get("calibration", "FETCH_SIZE + WRITE_SIZE per dispatch, separate rocprofv3 --pmc passes")
            traffic_step = pm.get("bytes_per_step")
    except Exception:
        pass
    if D["bound"] == "mfma":
        r_ach, r_peak, r_unit = D["achieved"], D["peak"], D["unit"]
    elif D["bound"] == "hbm":
        r_ach, r_peak, r_unit = D["achieved"], D["peak"], D["unit"]
    else:   # a vector kernel: price it in TFLOP/s-equivalents of the fp64 vector peak through its issue fraction
        r_ach, r_peak, r_unit = (D["frac"] or 0.0) * FP64_VALU_PEAK_TFLOPS, FP64_VALU_PEAK_TFLOPS, "TFLOP/s (issue fraction x fp64 vector peak)"
    ub = None
    try:
        ub = json.load(open(os.path.join(_ROOT, "profiles", "r03_ubench.json")))
    except Exception:
        pass
    # the step as a whole: every class's fraction of the roof that binds IT, weighted by its time (the classes of cs_column_profile)
    tw_num = sum((kern[k_]["frac"] or 0.0) * prof.get(cls[k_], 0.0) for k_ in cls)
    tw_den = sum(prof.get(cls[k_], 0.0) for k_ in cls)
    if "k_cheb_apply_mfma" in kern:
        tw_num += kern["k_cheb_apply_mfma"]["frac"] * prof["apply"]
        tw_den += prof["apply"]
    roofline = dict(bound=D["bound"], kernel=D.get("kernel", dom), achieved=r_ach, peak=r_peak,
                    unit=r_unit, frac=r_ach / r_peak if r_peak else None, traffic=traffic, traffic_note=traffic_note,
                    traffic_bytes_per_step=traffic_step, launches_per_step=D["launches_per_step"], avg_launch_ms=D["ms_per_launch"],
                    algorithmic_flops_per_launch=D.get("useful_flops_per_launch"), issued_flops_per_launch=D.get("issued_flops_per_launch"),
                    frac_issued=D.get("frac_issued"), algorithmic_bytes_per_launch=D.get("algorithmic_bytes_per_launch"),
                    note=("useful flops only: 2 x series terms for every (point | node, line, state) the reference sums there (inside the cut-off, "
                          "outside the core radius, real states); masked columns, padded states and step fill are in issued_flops.  The fp64 "
                          "matrix path of MI355X has the vector unit's peak (78.6 TFLOP/s) and shares its pipe; measured_rates holds what "
                          "tools/ubench sustains"),
                    measured_rates=ub, kernels=kern, kernel_ms=prof, reference_pair_evals=cnt["pair_evals"], interp_levels=work["levels"],
                    whole_step=dict(time_weighted_frac=(tw_num / tw_den) if tw_den > 0 else None,
                                    time_weighted_note="sum over kernel classes of (fraction of the roof that binds the class) x (its time), over the sum of the times",
                                    algorithmic_bytes=24.0 * col.nnu * col.nl + 58.0 * lines_total + 8.0 * col.nnu,
                                    hbm_frac=(24.0 * col.nnu * col.nl + 58.0 * lines_total + 8.0 * col.nnu) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    note="SURVEY 8d: 24 B per spectral point + 58 B per line + 8 B per nu over the whole step, against 8 TB/s -- "
                                         "small by construction: the path is fp64-issue-bound"))

    # the drop-in entry point (cs_fluxes_discretized: host pointers in, host arrays out, what the Julia method calls per
    # radiate!), PCIe-inclusive -- reported beside ms_per_step, never as `value`
    host_ptr = None
    if rank == 0 and N == 1 and not args.emulate_shard and not args.nu_range and not col.baked and not col.U.cia:
        from clearsky_jl_amd.core import _fluxes_discretized
        d = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                      theta_s=cfg["theta_s"], ctx=ctx, _setup=False)
        ctx2 = cs.Context(dev)      # a context of its own: first call = full setup, later calls re-use the resident column
        ctx2.set_precision(args.precision, args.far_s)      # ... with every setting of the run beside it
        ctx2.set_interp(not args.no_interp)
        ctx2.set_interp_plan(first_level=args.interp_first_level)
        ctx2.set_matrix_cores(0 if args.no_matrix_nodes else args.matrix_cores)
        ctx2.set_merge(not args.no_merge)
        for kv in filter(None, args.tune.split(",")):
            ctx2.set_tuning(*map(int, kv.split("=")))
        d.ctx = ctx2
        d.slots = np.array([ctx2.slot_of(g_.sl) for g_ in d.gases], dtype=np.int32)
        call_ms = []
        def timed(*bufs, reps=1):
            t1 = time.perf_counter()
            for _ in range(reps):
                t2 = time.perf_counter()
                Fq = _fluxes_discretized(d, *bufs)
                call_ms.append(round((time.perf_counter() - t2) * 1e3, 3))
            return (time.perf_counter() - t1) * 1e3 / reps, Fq
        import gc
        gc.collect()
        gc.disable()      # (a collection of the interpreter landing inside one of five 2 ms calls is not the library's time)
        first, _ = timed(None, None, None)
        timed(None, None, None)
        rep_f, Fq = timed(None, None, None, reps=5)
        tau_h = np.zeros((d.nl, d.nnu), order="F"); Mu_h = np.zeros((d.np, d.nnu), order="F"); Md_h = np.zeros((d.np, d.nnu), order="F")
        timed(tau_h, Mu_h, Md_h)    # (changes want_tau/want_M: one more setup)
        rep_all, _ = timed(tau_h, Mu_h, Md_h, reps=3)
        gc.enable()
        host_ptr = dict(first_call_ms=first, repeat_band_fluxes_ms=rep_f, repeat_with_tau_M_ms=rep_all,
                        radiate_bands_ms=rep_f, radiate_full_ms=rep_all,    # radiate!(F, HIPDiscretized(fluxpack=:bands | :full), ...): what heating! pays per step

                        d2h_bytes_with_tau_M=int(tau_h.nbytes + Mu_h.nbytes + Md_h.nbytes), olr_wm2=float(Fq[0][0]), call_ms=call_ms)
        del tau_h, Mu_h, Md_h
        ctx2.close()

    cpu = None
    if rank == 0 and N == 1 and not args.no_cpu:
        from oracle import oracle as O
        O.use_native_build()
        cia_data = [cs.readcia(x.x.filename) for x in col.U.cia] if col.U.cia else []
        # calibrate on a thin sample, then size the sample for roughly 15 s of CPU work
        def run_cpu(stride, want_sigma=False):
            sub = np.ascontiguousarray(nu[::stride])
            t1 = time.perf_counter()
            extra = None
            if cia_data:   # CIA continuum of the CPU port: numpy restatement, evaluated at the nodes like the CIA functor
                extra = np.zeros((col.K, len(sub)))
                for k in range(col.K):
                    for ci, x in enumerate(col.U.cia):
                        extra[k] += O.cia_sigma(cia_data[ci], sub, col.Tk[k], col.Pk[k], col.cia_P1[ci, k], col.cia_P2[ci, k],
                                                extrapolate=x.x.extrapolate, singles=x.x.singles)
            ref = O.fluxes_discretized(sub, cfg["P"], cfg["g"], cfg["core"].nlobatto, col.Tn, col.mun, col.Tlev,
                                       [g.sl for g in col.gases], [g.shape for g in col.gases], [g.dnu_cut for g in col.gases],
                                       col.conc, sigma_extra=extra, theta_s=cfg["theta_s"], nstream=cfg["core"].nstream,
                                       want_sigma=want_sigma)
            return sub, ref, time.perf_counter() - t1
        if args.cpu_stride > 0:
            stride = args.cpu_stride
        else:
            sub, ref, tc = run_cpu(64)
            stride = int(min(max(1, round(tc * 64 / 15.0)), 64))
        sub, ref, tc = run_cpu(stride, want_sigma=(stride == 1))
        cpu = dict(value=len(sub) * nl / tc, unit="spectral-points/s", cores=O.num_threads(), kind="port",
                   sample=f"every {stride}th wavenumber of the same column ({len(sub)} x {nl} points, {tc:.1f} s)",
                   olr_sample=float(ref["Fup"][0]),
                   olr_abs_err_wm2=(abs(olr - float(ref["Fup"][0])) if stride == 1 else None))
        if stride == 1:
            # the whole grid against the CPU port, EVERY element of what radiate! returns (fluxes.jl:357-383: tau, M+, M-, F+, F-) and of
            # the node cross-sections behind them: the step that was timed above, fetched now
            col.run(stream)
            torch.cuda.synchronize()
            tau_g = np.zeros((col.nl, col.nnu), order="F"); Mu_g = np.zeros((col.np, col.nnu), order="F"); Md_g = np.zeros((col.np, col.nnu), order="F")
            Fu_g, Fd_g = col.fetch(tau_g, Mu_g, Md_g)
            sig_g = col.sigma_nodes()
            sig_c = ref["sigma"]
            pos = sig_c > 0
            Mmax = float(max(np.max(ref["Mup"]), np.max(ref["Mdn"])))
            Fmax = float(np.max(ref["Fup"]))
            cpu.update(
                max_rel_sigma=float(np.max(np.abs(sig_g[pos] - sig_c[pos]) / sig_c[pos])) if pos.any() else 0.0,
                sigma_zero_where_cpu_zero=bool(np.all(sig_g[~pos] == 0.0)),
                max_rel_tau=float(np.max(np.abs(tau_g - ref["tau"]) / ref["tau"])),
                max_abs_M_over_max=float(max(np.max(np.abs(Mu_g - ref["Mup"])), np.max(np.abs(Md_g - ref["Mdn"]))) / Mmax),
                max_rel_F=float(max(np.max(np.abs(Fu_g - ref["Fup"])), np.max(np.abs(Fd_g - ref["Fdn"]))) / Fmax),
                max_rel_Fup=float(np.max(np.abs(Fu_g - ref["Fup"]) / ref["Fup"])),
                accuracy_note=("GPU vs CPU port on the full grid, every element: sigma [K, nnu] relative (where the CPU value is "
                               "non-zero), tau [nl, nnu] relative, M+ and M- [np, nnu] absolute over the column maximum, F+ and F- [np] "
                               "over max F+ (F-[1] = 0 at the top of the atmosphere), F+ also element-wise relative"))
            del tau_g, Mu_g, Md_g, sig_g, sig_c

    # The same column cut for 8 and 4 GPUs, one shard at a time on THIS card after the headline (the driver's scaling run needs an
    # 8-GPU node; this block puts the per-shard step under the driver's clock on a 1-GPU box too): shards 0, 3, 7 of 8 and 1 of 4 of the
    # partition re-cut from measured times, each timed like the headline (same stream, same step, fence on both sides).  A PROJECTION of
    # strong-scaling efficiency before the all-reduce, (headline ms / n) / shard ms -- not a measurement on n GPUs.
    emu = None
    if (rank == 0 and N == 1 and not args.emulate_shard and not args.nu_range and not args.no_emulated_shards and args.config == "C3"
            and args.nnu is None):
        emu = dict(note="one shard at a time on one GPU, after the headline, same context and stream; projected_efficiency = "
                        "(headline ms_per_step / n) / shard ms_per_step, before the band-flux all-reduce", shards=[])
        try:
            cuts = {}
            for n_ in (8, 4):
                base_ = W.balanced_ranges(nu, cfg["absorbers"], n_)
                times_ = []
                for r_i in range(n_):
                    tmp = make_column(base_[r_i])
                    times_.append(shard_ms(tmp, reps=8, batches=3))
                    del tmp
                cuts[n_] = cs.rebalance_ranges(nu, lines_pos, base_, times_, fixed_time=0.3 * min(times_))
            for r_, n_ in ((0, 8), (3, 8), (7, 8), (1, 4)):
                col = make_column(cuts[n_][r_])
                col.set_flux_dst(F.data_ptr())
                col.sync()
                for _ in range(args.warmup):
                    step()
                fence()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                fence()
                sm = (time.perf_counter() - t1) / args.steps * 1e3
                emu["shards"].append(dict(shard=f"{r_}/{n_}", nu_range=list(cuts[n_][r_]), ms_per_step=sm, launches_per_step=int(col.info()["launches"]),
                                          projected_efficiency=ms / n_ / sm))
            e8 = [x["projected_efficiency"] for x in emu["shards"] if x["shard"].endswith("/8")]
            emu["projected_efficiency_8gpu"] = min(e8)
            emu["projected_efficiency_4gpu"] = [x["projected_efficiency"] for x in emu["shards"] if x["shard"].endswith("/4")][0]
        except Exception as exc:
            emu["error"] = repr(exc)

    if rank == 0:
        out = dict(metric="spectral-points/s (nu x layers), whole-column LBL flux evaluation", value=value,
                   unit="spectral-points/s", n_gpus=N, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
                   higher_is_better=True, scaling="strong", vs_baseline=None,
                   dtype="f64" if args.precision == "fp64" else f"f64 with f32 far wings (x^2 >= {args.far_s:g})", data="synthetic",
                   config=dict(workload=f"{cfg['name']}: {'+'.join(g.formula for g in col.gases)}{' + CIA' if col.U.cia else ''} column, "
                                        f"{len(nu)} wavenumbers x {nl} layers, {args.shape}, {cfg['lines_kind']} lines "
                                        f"({lines_total} total), Discretized(nstream=5,nlobatto=2)",
                               nnu=len(nu), layers=nl, lines=lines_total, parallelism=f"nu-shard x{N}"),
                   collective=(f"{args.dist_backend} all_reduce of {2 * col.np} doubles per step" if use_dist else None),
                   olr_wm2=olr, setup_ms=setup_ms, launches_per_step=int(info["launches"]), launch_groups=int(info["groups"]), host_pointer_ms=host_ptr, kernel_source_sha16=source_stamp(),
                   partition=partition, emulated_shards=emu, roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
