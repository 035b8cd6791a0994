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
import socket
import subprocess
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

# flops of one far-wing Voigt term in k_voigt_far: 14 fp64 VALU instructions (ISA count, mode 0/1 mix), FMAs = 2 flops
FLOPS_PER_PAIR = 24.0
# VALU instructions per (lane, line) of each loop body, counted in the gfx950 ISA of k_voigt_far / k_cheb_nodes (hipcc -S, loops
# unrolled by 4: 52/68/64/80/96/148 per four lines; profiles/r02_notes.md).  The 2-/3-/4-term far-wing bodies are the same
# instructions in both kernels; "_cut" adds the cut-off compare + select (and |dnu|).
VALU_PER_LINE = dict(t2=13, t2_cut=17, t3=16, t3_cut=20, t4=20, t4_cut=24, near_zone=37)
VALU_ISSUE_PEAK = 256 * 4 * 16 * 2.4e9     # fp64-rate lane-instructions per second: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def source_stamp():
    """sha256 (16 hex digits) of the kernel sources the loaded library was built from: PMC traffic collected offline is only
    quoted while it belongs to these exact kernels (profiles/pmc_traffic.json carries the stamp of the build it was measured on)."""
    h = hashlib.sha256()
    csrc = os.path.join(_ROOT, "clearsky.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


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
    ap.add_argument("--cpu-stride", type=int, default=0, help="cpu_baseline evaluates every n-th wavenumber (0 = size the sample for ~15 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as fresh child processes, BEFORE anything
        # in this process touches the GPU (no exec of a GPU-initialised process), and exit with the launcher's code
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
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
    if N > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.dist_backend)

    cfg = W.config(args.config, nnu=args.nnu, lines_kind=args.lines, shape=args.shape)
    nu, nl = cfg["nu"], cfg["nl"]
    ranges = W.balanced_ranges(nu, cfg["absorbers"], N)
    if args.emulate_shard:
        r_, n_ = map(int, args.emulate_shard.split("/"))
        ranges = [W.balanced_ranges(nu, cfg["absorbers"], n_)[r_]]
    ctx = cs.Context(dev)
    ctx.set_precision(args.precision, args.far_s)
    ctx.set_interp(not args.no_interp)
    ctx.set_interp_plan(first_level=args.interp_first_level)
    ctx.set_matrix_cores(0 if args.no_matrix_nodes else args.matrix_cores)
    ctx.set_merge(not args.no_merge)
    for kv in filter(None, args.tune.split(",")):
        ctx.set_tuning(*map(int, kv.split("=")))
    t_setup = time.perf_counter()
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    theta_s=cfg["theta_s"], want_tau=True, want_M=True, nu_range=ranges[rank], ctx=ctx)
    col.sync()
    setup_ms = (time.perf_counter() - t_setup) * 1e3   # one-off: closures, windows, interpolation matrices, workspaces, uploads
    # one explicit torch stream carries the kernels, the D2D copy of the band fluxes and the collective, so they are
    # ordered by the stream (torch's default stream has handle 0, which the C ABI reads as "use the context's stream")
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    F = torch.zeros(2 * col.np, dtype=torch.float64, device=f"cuda:{dev}")

    def step():
        col.run(stream)
        col.flux_to(F.data_ptr(), stream)
        if N > 1:
            if args.dist_backend == "nccl":
                dist.all_reduce(F)   # RCCL over xGMI: 2*np doubles, the only collective of the path
            else:
                Fc = F.cpu()
                dist.all_reduce(Fc)
                F.copy_(Fc)

    def fence():
        torch.cuda.synchronize()
        if N > 1:
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
    if N > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    points = len(nu) * nl
    value = points * args.steps / dt
    Fh = F.cpu().numpy()
    olr = float(Fh[0])

    # per-kernel HIP-event timing on the launch stream (rank-local) and the roofline of the dominant kernel
    prof = col.profile(reps=max(3, min(10, args.steps)), stream=stream)
    cnt = col.counts()
    info = col.info()
    ngas = max(int(info["groups"]), 1)     # launch sets per step: the gases of a column that share a cut-off run as ONE merged table
    K = col.K
    lines_total = sum(len(g.sl.nu) for g in col.gases)
    # k_voigt_far (one launch per gas), the longest kernel.  Algorithmic HBM bytes per launch: sigma written
    # (8 B per (nu, node)) and re-read when it accumulates (a later gas, or onto the interpolated far wings) + nu, and for the states
    # whose window core is not k_voigt_sub's (K_far of K on average): the 32-B record of every (node, line) read once + the 8-B near-line
    # index words handed to k_voigt_near per (nu, node).
    work = col.work()
    interp_on = work["levels"] > 0
    nt64 = (col.nnu + 63) // 64
    K_far = K - work.get("core_tile_states", 0) / max(ngas, 1) / nt64
    far_bytes = [32 * K_far * lines_total / ngas + 8 * col.nnu * K * (2 if (gi > 0 or interp_on) else 1) + 8 * col.nnu * K_far + 8 * col.nnu
                 for gi in range(ngas)]
    far_ms = prof["far"] / max(ngas, 1)
    alg = float(np.mean(far_bytes)) if far_bytes else 0.0
    achieved = alg / (far_ms * 1e-3) / 1e9 if far_ms > 0 else 0.0
    # PMC traffic cannot be collected inside this process (rocprofv3 --pmc wraps the program, FETCH_SIZE and WRITE_SIZE in
    # separate passes: tools/profile.sh writes profiles/pmc_traffic.json).  It is quoted only when that file was measured on the
    # kernels loaded now (same source stamp) and on this exact workload; otherwise null -- never a stale number.
    traffic, traffic_note = None, "no PMC profile of this build/workload under profiles/"
    pm, pm_dominant = None, None
    try:
        pm = json.load(open(os.path.join(_ROOT, "profiles", "pmc_traffic.json")))
        default_wl = args.nnu is None and args.lines is None and args.shape == "voigt" and N == 1 and interp_on and args.precision == "fp64" and not args.emulate_shard
        if pm.get("source_sha16") != source_stamp():
            traffic_note = f"profiles/pmc_traffic.json belongs to build {pm.get('source_sha16')}, loaded kernels are {source_stamp()}"
        elif pm.get("config") != args.config or not default_wl:
            traffic_note = "profiles/pmc_traffic.json was measured on another workload"
        else:
            kk = pm["kernels"][pm["dominant"]]
            pm_dominant = pm["dominant"].split("<")[0]
            traffic = (kk["FETCH_SIZE_KB"] + kk["WRITE_SIZE_KB"]) * 1024.0
            traffic_note = pm.get("calibration", "")
    except Exception:
        pass
    # fp64 VALU view of the three line kernels: evaluations actually issued (per-point ones count all 64 lanes of a wave, node
    # ones 64 nodes per (interval, line)) x 24 flops, over their time.  `reference_pair_evals` is what surf! evaluates.
    # (node_evals counts every node sum; node_evals_matrix / direct_evals_matrix of them / besides direct_evals run on the matrix cores)
    mx_triples = work.get("node_evals_matrix", 0) + work.get("direct_evals_matrix", 0)
    evals = work["direct_evals"] + work["node_evals"] - work.get("node_evals_matrix", 0) + work.get("sub_evals", 0)
    flops = evals * FLOPS_PER_PAIR
    line_ms = prof["nodes"] + prof["far"] + prof["near"] + prof.get("sub", 0.0)
    mx_ms = prof.get("nodes_mx", 0.0) + prof.get("far_mx", 0.0)
    # matrix cores: 4 (or 3, or 8) terms x (multiply + add) per (node | point, line, state); v_mfma_f64_16x16x4 = 2048 flop.  Spec peak of the fp64
    # matrix path = the fp64 vector rate (78.6 TFLOP/s); tools/ubench/mfma_f64_rate.hip sustains 47, and a matrix and a vector
    # kernel launched side by side on two streams take the sum of their times (tools/ubench/sep_nodes.hip): one fp64 pipe, two ways in
    mx_flops = 8.0 * mx_triples - 2.0 * work.get("matrix_evals_3term", 0) + 8.0 * work.get("matrix_evals_8term", 0)   # (3 terms: 6 flop; 8 terms: 16)
    matrix_fp64 = dict(triples=mx_triples, triples_3term=work.get("matrix_evals_3term", 0), triples_8term=work.get("matrix_evals_8term", 0), flops=mx_flops, ms=mx_ms,
                       achieved=mx_flops / (mx_ms * 1e-3) / 1e12 if mx_ms > 0 else 0.0, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s",
                       frac=mx_flops / (mx_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS if mx_ms > 0 else 0.0,
                       measured_mfma_f64_rate=47.0, kernels="k_cheb_nodes_mx + k_voigt_edge_mx")
    # measured instruction mix: lane-instructions each far-wing kernel issues = sum over its loop bodies of (lines x 64 lanes) x VALU
    # instructions per line (ISA), over its HIP-event time, against the fp64-rate issue peak (all 64 lanes of a wave count, also
    # those the cut-off predicate masks)
    db, nb = work.get("direct_by_body", {}), work.get("node_by_body", {})
    far_instr = sum(db.get(b, 0) * VALU_PER_LINE[b] for b in db)
    sub_instr = work.get("sub_evals", 0) * VALU_PER_LINE["near_zone"]
    node_instr = sum(nb.get(b, 0) * VALU_PER_LINE[b] for b in nb)
    valu_issue = dict(unit="fraction of the fp64-rate VALU issue peak (256 CU x 4 SIMD x 16 lanes x 2.4 GHz)", valu_per_line=VALU_PER_LINE,
                      k_voigt_far=dict(lane_instr=far_instr, ms=prof["far"], frac=(far_instr / (prof["far"] * 1e-3) / VALU_ISSUE_PEAK) if prof["far"] > 0 else None,
                                       lines_x_lanes_by_body=db),
                      k_voigt_sub=dict(lane_instr=sub_instr, ms=prof.get("sub", 0.0), evals=work.get("sub_evals", 0),
                                       frac=(sub_instr / (prof["sub"] * 1e-3) / VALU_ISSUE_PEAK) if prof.get("sub", 0.0) > 0 else None),
                      k_cheb_nodes=dict(lane_instr=node_instr, ms=prof["nodes"], frac=(node_instr / (prof["nodes"] * 1e-3) / VALU_ISSUE_PEAK) if prof["nodes"] > 0 else None,
                                        lines_x_nodes_by_body=nb))
    # the dominant kernel = the class with the largest time per step among the line kernels (all one launch per gas); its roof:
    # the fp64 matrix path for the two matrix-core kernels (flops of the series terms they sum), HBM for the vector kernels
    # (algorithmic bytes; their binding roof, the fp64 vector unit, is in valu_issue)
    n3n = work.get("node_evals_matrix_3term", 0)
    fl_nodes_mx = 8.0 * work.get("node_evals_matrix", 0) - 2.0 * n3n
    fl_edge_mx = 8.0 * work.get("direct_evals_matrix", 0) - 2.0 * (work.get("matrix_evals_3term", 0) - n3n) + 8.0 * work.get("matrix_evals_8term", 0)
    K_vec = K   # (k_cheb_nodes: the records of every state can be touched; F written once)
    nodes_bytes = (32 * K_vec * lines_total / ngas + 8 * 64 * work["intervals"] * K) if col.gases else 0.0
    cands = dict(k_voigt_far=("hbm", prof["far"], alg), k_cheb_nodes=("hbm", prof["nodes"], nodes_bytes),
                 k_voigt_edge_mx=("mfma", prof.get("far_mx", 0.0), fl_edge_mx / max(ngas, 1)),
                 k_cheb_nodes_mx=("mfma", prof.get("nodes_mx", 0.0), fl_nodes_mx / max(ngas, 1)))
    dom = max(cands, key=lambda kname: cands[kname][1])
    bound, dom_ms, per_launch = cands[dom]
    dom_ms /= max(ngas, 1)
    if bound == "mfma":
        r_ach, r_peak, r_unit = (per_launch / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0), FP64_VALU_PEAK_TFLOPS, "TFLOP/s"
    else:
        r_ach, r_peak, r_unit = (per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0), HBM_PEAK_GBS, "GB/s"
    if traffic is not None and pm_dominant != dom:
        traffic, traffic_note = None, f"profiles/pmc_traffic.json names {pm_dominant} as the dominant kernel, this run {dom}"
    elif traffic is not None:
        kk = pm["kernels"][[k_ for k_ in pm["kernels"] if k_.split("<")[0] == dom][0]]
        traffic = (kk["FETCH_SIZE_KB"] + kk["WRITE_SIZE_KB"]) * 1024.0
    roofline = dict(bound=bound, kernel=dom, achieved=r_ach, peak=r_peak, unit=r_unit,
                    frac=r_ach / r_peak, traffic=traffic, traffic_note=traffic_note, launches_per_step=ngas, avg_launch_ms=dom_ms,
                    algorithmic_flops_per_launch=per_launch if bound == "mfma" else None,
                    algorithmic_bytes_per_launch=per_launch if bound == "hbm" else None,
                    note=("series terms x (multiply + add) per (nu | node, line, state) summed as v_mfma_f64_16x16x4 products; the fp64 matrix path of "
                          "MI355X has the vector unit's peak (78.6 TFLOP/s; tools/ubench/mfma_f64_rate.hip sustains 47) and shares its pipe"
                          if bound == "mfma" else "elementwise fp64 accumulate over (nu,line) pairs: VALU-bound by construction, see valu_issue"),
                    k_voigt_far_hbm=dict(achieved=achieved, frac=achieved / HBM_PEAK_GBS, unit="GB/s", avg_launch_ms=far_ms, algorithmic_bytes_per_launch=alg),
                    valu_fp64=dict(achieved=flops / (line_ms * 1e-3) / 1e12 if line_ms > 0 else 0.0, peak=FP64_VALU_PEAK_TFLOPS,
                                   unit="TFLOP/s", frac=flops / (line_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS if line_ms > 0 else 0.0,
                                   evals_issued=evals, direct_evals=work["direct_evals"], node_evals=work["node_evals"],
                                   reference_pair_evals=cnt["pair_evals"], flops_per_eval=FLOPS_PER_PAIR,
                                   kernels="k_cheb_nodes + k_voigt_far + k_voigt_sub + k_voigt_near"),
                    valu_issue=valu_issue, matrix_fp64=matrix_fp64,
                    interp_levels=work["levels"], kernel_ms=prof)

    # the drop-in entry point (cs_fluxes_discretized: host pointers in, host arrays out, what the Julia method calls per
    # radiate!), PCIe-inclusive -- reported beside ms_per_step, never as `value`
    host_ptr = None
    if rank == 0 and N == 1 and not args.emulate_shard and not col.baked and not col.U.cia:
        from clearsky_jl_amd.core import _fluxes_discretized
        d = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                      theta_s=cfg["theta_s"], ctx=ctx, _setup=False)
        ctx2 = cs.Context(dev)      # a context of its own: first call = full setup, later calls re-use the resident column
        ctx2.set_precision(args.precision, args.far_s)
        ctx2.set_interp(not args.no_interp)
        d.ctx = ctx2
        d.slots = np.array([ctx2.slot_of(g_.sl) for g_ in d.gases], dtype=np.int32)
        def timed(*bufs, reps=1):
            t1 = time.perf_counter()
            for _ in range(reps):
                Fq = _fluxes_discretized(d, *bufs)
            return (time.perf_counter() - t1) * 1e3 / reps, Fq
        first, _ = timed(None, None, None)
        rep_f, Fq = timed(None, None, None, reps=5)
        tau_h = np.zeros((d.nl, d.nnu), order="F"); Mu_h = np.zeros((d.np, d.nnu), order="F"); Md_h = np.zeros((d.np, d.nnu), order="F")
        timed(tau_h, Mu_h, Md_h)    # (changes want_tau/want_M: one more setup)
        rep_all, _ = timed(tau_h, Mu_h, Md_h, reps=3)
        host_ptr = dict(first_call_ms=first, repeat_band_fluxes_ms=rep_f, repeat_with_tau_M_ms=rep_all,
                        d2h_bytes_with_tau_M=int(tau_h.nbytes + Mu_h.nbytes + Md_h.nbytes), olr_wm2=float(Fq[0][0]))
        del tau_h, Mu_h, Md_h
        ctx2.close()

    cpu = None
    if rank == 0 and N == 1 and not args.no_cpu:
        from oracle import oracle as O
        O.use_native_build()
        cia_data = [cs.readcia(x.x.filename) for x in col.U.cia] if col.U.cia else []
        # calibrate on a thin sample, then size the sample for roughly 15 s of CPU work
        def run_cpu(stride):
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
                                       col.conc, sigma_extra=extra, theta_s=cfg["theta_s"], nstream=cfg["core"].nstream)
            return sub, ref, time.perf_counter() - t1
        if args.cpu_stride > 0:
            stride = args.cpu_stride
        else:
            sub, ref, tc = run_cpu(64)
            stride = int(min(max(1, round(tc * 64 / 15.0)), 64))
        sub, ref, tc = run_cpu(stride)
        cpu = dict(value=len(sub) * nl / tc, unit="spectral-points/s", cores=O.num_threads(), kind="port",
                   sample=f"every {stride}th wavenumber of the same column ({len(sub)} x {nl} points, {tc:.1f} s)",
                   olr_sample=float(ref["Fup"][0]),
                   olr_abs_err_wm2=(abs(olr - float(ref["Fup"][0])) if stride == 1 else None))

    if rank == 0:
        out = dict(metric="spectral-points/s (nu x layers), whole-column LBL flux evaluation", value=value,
                   unit="spectral-points/s", n_gpus=N, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
                   higher_is_better=True, scaling="strong", vs_baseline=None,
                   dtype="f64" if args.precision == "fp64" else f"f64 with f32 far wings (x^2 >= {args.far_s:g})", data="synthetic",
                   config=dict(workload=f"{cfg['name']}: {'+'.join(g.formula for g in col.gases)}{' + CIA' if col.U.cia else ''} column, "
                                        f"{len(nu)} wavenumbers x {nl} layers, {args.shape}, {cfg['lines_kind']} lines "
                                        f"({lines_total} total), Discretized(nstream=5,nlobatto=2)",
                               nnu=len(nu), layers=nl, lines=lines_total, parallelism=f"nu-shard x{N}"),
                   olr_wm2=olr, setup_ms=setup_ms, launches_per_step=int(info["launches"]), launch_groups=int(info["groups"]), host_pointer_ms=host_ptr, kernel_source_sha16=source_stamp(),
                   roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out))
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
