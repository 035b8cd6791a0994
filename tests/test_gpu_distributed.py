"""The N > 1 path on real kernels: two ranks (gloo rendezvous, both on the box's one card) each run the HIP column on their
own work-balanced nu shard -- `Column(nu_range=...)`, global trapezoid weights -- and combine the band fluxes with ONE
all-reduce of 2*np doubles; the sum must be the oracle's full column.  Also: `bench.py --gpus 2` started WITHOUT a launcher
spawns its two ranks itself and prints one JSON line for the whole job."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import clearsky_jl_amd as cs
    import workloads as W
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = W.config("C3", nnu=16000, nl=20, lines_kind="fixture", nu_span=(500.0, 900.0))     # 0.025 cm^-1: three interval levels
    rng = W.balanced_ranges(cfg["nu"], cfg["absorbers"], world)[rank]
    ctx = cs.Context(0)
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    theta_s=cfg["theta_s"], want_tau=False, want_M=False, nu_range=rng, ctx=ctx)
    col.run()
    F = torch.from_numpy(np.concatenate(col.fetch()))
    dist.all_reduce(F)                                   # the path's only collective (RCCL in bench.py; gloo here)
    if rank == 0:
        from oracle import oracle as O
        r = O.fluxes_discretized(cfg["nu"], cfg["P"], cfg["g"], 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 2,
                                 [25.0] * 2, col.conc, theta_s=cfg["theta_s"], nstream=5)
        ref = np.concatenate([r["Fup"], r["Fdn"]])
        q.put((float(np.max(np.abs(F.numpy() - ref)) / ref.max()), rng))
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_real_columns_reduce_to_the_oracle():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, rng = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert err < 1e-11
    assert rng[0] == 0 and rng[1] % 64 == 0            # shard edges sit on tile boundaries


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK/WORLD_SIZE in the environment: two child ranks, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--config", "C2",
                          "--steps", "3", "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "nu-shard x2"
    # a rank count that does not match the launcher's is an error, not a silent 1-GPU run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"], env=env2, capture_output=True,
                         text=True, timeout=120)
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr


def test_multi_context_entry_point_vs_oracle(cs, O, lines):
    """cs_fluxes_discretized_multi, the product's N-GPU form of B3 (radiate! with `ngpu` devices behind it): N contexts -- here two and
    three on the box's one card -- each take a cost-balanced wavenumber range; tau, M+, M- land in the caller's arrays range by range,
    band fluxes are added on the host in context order.  Against the oracle's whole column, against the one-context call bit for
    bit where the arithmetic is per wavenumber (tau, M+, M-), and repeatable bit for bit."""
    import workloads as W
    nu = np.linspace(550.0, 800.0, 12000)
    P = cs.pressuregrid(10.0, 1e5, 13)
    T = W.earth_temperature(P)
    gases = [cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu), cs.GrayGas(3e-28, nu)]
    one = cs.Context(0)
    F1 = cs.radiate(P, 9.8, T, 0.029, 0.3, 0.2, *gases, core=cs.Discretized(5, 3), ctx=one)
    col = cs.Column(P, 9.8, T, 0.029, 0.3, 0.2, *gases, core=cs.Discretized(5, 3), ctx=one, _setup=False)
    ref = O.fluxes_discretized(nu, P, 9.8, 3, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 2, [25.0] * 2, col.conc,
                               sigma_gray=col.sigma_gray, S_toa=col.S_toa, albedo=col.albedo)
    for n in (2, 3):
        mc = cs.MultiContext([0] * n)
        F = cs.radiate(P, 9.8, T, 0.029, 0.3, 0.2, *gases, core=cs.Discretized(5, 3), ctx=mc)
        assert np.max(np.abs(F.tau - ref["tau"]) / ref["tau"]) < 1e-11
        sm = ref["Mup"].max()
        assert np.max(np.abs(F.Mup - ref["Mup"])) < 1e-11 * sm and np.max(np.abs(F.Mdn - ref["Mdn"])) < 1e-11 * sm
        assert np.max(np.abs(F.Fup - ref["Fup"])) < 1e-11 * ref["Fup"].max() and np.max(np.abs(F.Fdn - ref["Fdn"])) < 1e-11 * ref["Fup"].max()
        # per-wavenumber results do not depend on how the grid was cut beyond the rounding of the far-wing interpolation
        assert np.max(np.abs(F.tau - F1.tau) / F1.tau) < 5e-13
        assert np.max(np.abs(F.Fup - F1.Fup)) < 1e-12 * F1.Fup.max()
        # a second call re-uses the resident shards (same grid) with a new temperature profile; and is bitwise repeatable
        G1 = cs.radiate(P, 9.8, T + 3.0, 0.029, 0.3, 0.2, *gases, core=cs.Discretized(5, 3), ctx=mc)
        G2 = cs.radiate(P, 9.8, T + 3.0, 0.029, 0.3, 0.2, *gases, core=cs.Discretized(5, 3), ctx=mc)
        assert np.array_equal(G1.Fup, G2.Fup) and np.array_equal(G1.Mup, G2.Mup) and not np.array_equal(G1.Fup, F.Fup)
        Fb = cs.fluxes(P, 9.8, T + 3.0, 0.029, 0.3, 0.2, *gases, core=cs.Discretized(5, 3), ctx=mc)      # band fluxes only (NULL tau, M)
        assert np.max(np.abs(Fb[0] - G1.Fup)) < 1e-13 * G1.Fup.max()
        mc.close()
    # contexts that do not hold the same tables are refused
    import ctypes as C
    from clearsky_jl_amd._lib import lib
    a, b = cs.Context(0), cs.Context(0)
    a.slot_of(lines("CO2"))
    h = (C.c_void_p * 2)(a.handle.value, b.handle.value)
    z = np.zeros(len(P))
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    slots = (C.c_int * 1)(0)
    Tn = np.asfortranarray(col.Tn).ravel(order="F").copy()
    rc = lib().cs_fluxes_discretized_multi(h, 2, len(nu), dp(nu), len(P), dp(P), 9.8, 3, dp(Tn), dp(Tn), dp(col.Tlev), 1, slots, None, None,
                                           dp(np.full(col.K, 4e-4)), 0.0, None, None, None, 0.841, 5, None, None, None, dp(z), dp(z.copy()))
    assert rc == -1 and b"same table on every context" in lib().cs_last_error()
    a.close(); b.close(); one.close()


def test_bench_rccl_path_single_rank():
    """The RCCL line of bench.py (init_process_group("nccl", device_id=...) + all_reduce of the band fluxes on the bench stream) with
    the one rank a one-GPU box can hold: the collective the N > 1 job runs per step executes here too, and must leave the fluxes as
    they are."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = {}
    for flag in ([], ["--force-dist", "--dist-backend", "nccl"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C2", "--steps", "5", "--warmup", "2", "--no-cpu"] + flag,
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        res[bool(flag)] = d
    assert res[True]["collective"].startswith("nccl") and res[False]["collective"] is None
    assert res[True]["olr_wm2"] == res[False]["olr_wm2"]


def test_multi_context_partition_recut_from_measured_times(cs, O, lines):
    """The product's N-GPU call re-cuts its partition once, inside the first call on a grid, from what the ranges were measured to take
    (cs_rebalance_ranges; on the box's one card the times are those of contexts taking turns -- cs_set_tuning key 15 | 32 forces the
    calibration there, which exercises the path: measure, re-cut, set the shards up again).  Results stay the oracle's; later calls keep
    the re-cut partition and are bitwise repeatable."""
    import workloads as W
    nu = np.linspace(550.0, 800.0, 16000)
    P = cs.pressuregrid(10.0, 1e5, 11)
    T = W.earth_temperature(P)
    gases = [cs.DirectGas(lines("H2O"), W.fC_h2o, nu), cs.DirectGas(lines("CO2"), 400e-6, nu)]
    col = cs.Column(P, 9.8, T, 0.029, 0.0, 0.0, *gases, core=cs.Discretized(5, 2), ctx=cs.Context(0), _setup=False)
    ref = O.fluxes_discretized(nu, P, 9.8, 2, col.Tn, col.mun, col.Tlev, [g.sl for g in col.gases], ["voigt"] * 2, [25.0] * 2, col.conc)
    mc = cs.MultiContext([0, 0, 0])
    mc.ctxs[0].set_tuning(15, 32)
    F = cs.radiate(P, 9.8, T, 0.029, 0.0, 0.0, *gases, core=cs.Discretized(5, 2), ctx=mc)
    assert np.max(np.abs(F.tau - ref["tau"]) / ref["tau"]) < 1e-11
    assert np.max(np.abs(F.Fup - ref["Fup"])) < 1e-11 * ref["Fup"].max()
    G = cs.radiate(P, 9.8, T, 0.029, 0.0, 0.0, *gases, core=cs.Discretized(5, 2), ctx=mc)
    assert np.array_equal(F.Fup, G.Fup) and np.array_equal(F.tau, G.tau) and np.array_equal(F.Mup, G.Mup)
    mc.close()


def test_bench_emulated_shard_with_calibrated_partition():
    """bench.py --emulate-shard r/N re-cuts the N-way partition from the N shards' measured times before its timed region (what every rank
    of an N-GPU run does through one all-reduce of N doubles at setup time) and reports both partitions"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "5", "--warmup", "2", "--no-cpu",
                          "--emulate-shard", "1/3"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    p = d["partition"]
    assert p is not None and "error" not in p, p
    assert len(p["model_ranges"]) == 3 and len(p["ranges"]) == 3 and len(p["model_shard_ms"]) == 3
    assert p["ranges"][0][0] == 0 and p["ranges"][-1][1] == d["config"]["nnu"]
    assert all(p["ranges"][i][1] == p["ranges"][i + 1][0] for i in range(2))
    off = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "5", "--warmup", "2", "--no-cpu",
                          "--emulate-shard", "1/3", "--no-calibrate"], env=env, capture_output=True, text=True, timeout=600)
    assert off.returncode == 0 and json.loads([l for l in off.stdout.splitlines() if l.startswith("{")][0])["partition"] is None


@pytest.mark.parametrize("fail", [None, "measure:1", "setup:0"])
def test_bench_calibration_is_collective(fail):
    """the N-rank partition calibration of bench.py.  Every rank reaches the same all-reduces whatever happened to it locally,
    and the re-cut partition is adopted only if EVERY rank measured its shard and set its new one up; a rank that fails (injected:
    CS_BENCH_FAIL_CALIBRATION=measure:<rank> | setup:<rank>) sends every rank back to the model's ranges -- no hang, no mixed partition.
    Rank 0 gathers the ranges the ranks actually run and refuses a result whose ranges do not tile the grid."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if fail:
        env["CS_BENCH_FAIL_CALIBRATION"] = fail
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--config", "C2",
                          "--steps", "3", "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    p = d["partition"]
    by_rank = p["ranges_by_rank"]
    assert len(by_rank) == 2 and by_rank[0][0] == 0 and by_rank[0][1] == by_rank[1][0] and by_rank[1][1] == d["config"]["nnu"]
    assert [list(r) for r in p["ranges"]] == by_rank
    if fail:
        assert p["calibrated"] is False and "error" in p and [list(r) for r in p["model_ranges"]] == by_rank
    else:
        assert "error" not in p and len(p["model_shard_ms"]) == 2 and all(t > 0 for t in p["model_shard_ms"])
    assert d["olr_wm2"] > 0
