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
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

# fp64 ops of one far-wing Voigt term in k_linesum (sub, mul, fma x? ... counted from the ISA: 17 VALU ops, FMAs = 2 flops)
FLOPS_PER_PAIR = 34.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--lines", default=None, help="synthetic | fixture")
    ap.add_argument("--nnu", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-stride", type=int, default=8, help="cpu_baseline evaluates every n-th wavenumber")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    N = world

    import torch
    import clearsky_jl_amd as cs
    from clearsky_jl_amd import workloads as W

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the HIP path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if N > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = W.config(args.config, nnu=args.nnu, lines_kind=args.lines)
    nu, nl = cfg["nu"], cfg["nl"]
    ranges = W.balanced_ranges(nu, cfg["absorbers"], N)
    ctx = cs.Context(local_rank)
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    theta_s=cfg["theta_s"], want_tau=True, want_M=True, nu_range=ranges[rank], ctx=ctx)
    stream = torch.cuda.current_stream().cuda_stream
    F = torch.zeros(2 * col.np, dtype=torch.float64, device=f"cuda:{local_rank}")

    def step():
        col.run(stream)
        col.flux_to(F.data_ptr(), stream)
        if N > 1:
            dist.all_reduce(F)   # RCCL over xGMI: 2*np doubles

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
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
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
    ngas = len(col.gases)
    K = col.K
    lines_total = sum(len(g.sl.nu) for g in col.gases)
    # algorithmic HBM bytes of the k_linesum launches of one evaluation (DESIGN.md "Kernels"):
    #   read nu (8 B/point/launch) + read the 32-B hot parameter record of every (node, line) once
    #   + write sigma (8 B per (nu,node)), + re-read it when a later gas accumulates
    ls_bytes = ngas * 8 * col.nnu + 32 * K * lines_total + 8 * col.nnu * K * (2 * ngas - 1)
    ls_ms = prof["linesum"]
    achieved = ls_bytes / (ls_ms * 1e-3) / 1e9 if ls_ms > 0 else 0.0
    roofline = dict(bound="hbm", kernel="k_linesum", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=None,
                    launches_per_step=ngas, avg_launch_ms=ls_ms / max(ngas, 1), algorithmic_bytes_per_step=ls_bytes,
                    valu_fp64=dict(achieved=cnt["pair_evals"] * FLOPS_PER_PAIR / (ls_ms * 1e-3) / 1e12 if ls_ms > 0 else 0.0,
                                   peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s",
                                   frac=cnt["pair_evals"] * FLOPS_PER_PAIR / (ls_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
                                   if ls_ms > 0 else 0.0, pair_evals=cnt["pair_evals"]),
                    kernel_ms=prof)

    cpu = None
    if rank == 0 and N == 1 and not args.no_cpu:
        from oracle import oracle as O
        O.use_native_build()
        stride = max(1, args.cpu_stride)
        sub = np.ascontiguousarray(nu[::stride])
        full = col   # node states (Tn, mun, Tlev, conc) do not depend on the wavenumber shard
        t1 = time.perf_counter()
        ref = O.fluxes_discretized(sub, cfg["P"], cfg["g"], cfg["core"].nlobatto, full.Tn, full.mun, full.Tlev,
                                   [g.sl for g in col.gases], [g.shape for g in col.gases], [g.dnu_cut for g in col.gases],
                                   full.conc, theta_s=cfg["theta_s"], nstream=cfg["core"].nstream)
        tc = time.perf_counter() - t1
        cpu = dict(value=len(sub) * nl / tc, unit="spectral-points/s", cores=O.num_threads(), kind="port",
                   sample=f"every {stride}th wavenumber of the same column ({len(sub)} x {nl} points, {tc:.1f} s)",
                   olr_sample=float(ref["Fup"][0]))

    if rank == 0:
        out = dict(metric="spectral-points/s (nu x layers), whole-column LBL flux evaluation", value=value,
                   unit="spectral-points/s", n_gpus=N, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
                   higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=f"{cfg['name']}: {'+'.join(g.formula for g in col.gases)} column, "
                                        f"{len(nu)} wavenumbers x {nl} layers, Voigt, {cfg['lines_kind']} lines "
                                        f"({lines_total} total), Discretized(nstream=5,nlobatto=2)",
                               nnu=len(nu), layers=nl, lines=lines_total, parallelism=f"nu-shard x{N}"),
                   olr_wm2=olr, roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out))
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
