"""world_size-2 gloo test of the multi-GPU decomposition on CPU: contiguous work-balanced nu shards,
global trapezoid weights, ONE all-reduce of 2*np doubles.  The per-shard compute is the oracle here (no GPU in this
container); on the GPU box the same decomposition is exercised by tests/test_gpu_parity.py::test_shards_add_up."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import clearsky_jl_amd as cs
    import workloads as W
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = W.config("C2", nnu=1200, nl=10)
    nu, P = cfg["nu"], cfg["P"]
    j0, j1 = W.balanced_ranges(nu, cfg["absorbers"], world)[rank]
    fT, fmu = cs.formprofile(P, cfg["T"]), cs.formprofile(P, cfg["mu"])
    Tn, mun = cs.lobattoevaluations(P, fT, fmu, 2)
    Tlev = np.array([fT(p) for p in P])
    gas = cfg["absorbers"][0]
    conc = np.full((1, len(P)), 400e-6)
    kw = dict(theta_s=0.841, nstream=5)
    part = O.fluxes_discretized(nu[j0:j1], P, 9.8, 2, Tn, mun, Tlev, [gas.sl], ["voigt"], [25.0], conc, **kw)
    w = cs.trapz_weights(nu)[j0:j1]
    F = torch.from_numpy(np.concatenate([part["Mup"] @ w, part["Mdn"] @ w]))
    dist.all_reduce(F)
    if rank == 0:
        full = O.fluxes_discretized(nu, P, 9.8, 2, Tn, mun, Tlev, [gas.sl], ["voigt"], [25.0], conc, **kw)
        ref = np.concatenate([full["Fup"], full["Fdn"]])
        q.put(float(np.max(np.abs(F.numpy() - ref) / np.maximum(np.abs(ref), 1e-6 * ref.max()))))
    dist.destroy_process_group()


def test_two_rank_shards_reduce_to_full_column():
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
    err = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-12


def test_balanced_ranges_partition_properties():
    """cs_balanced_ranges (the partition cs_fluxes_discretized_multi and bench.py cut the grid with; host only): contiguous, covering,
    non-empty ranges -- also with a dense line cluster at the top of a grid whose length is not a multiple of the 64-point tile
    (an interior edge used to round past the end) -- edges on tile boundaries where the grid is long enough, and about
    equal estimated cost."""
    import clearsky_jl_amd as cs
    rng = np.random.default_rng(5)
    nu = np.linspace(1.0, 100.0, 100037)
    cluster = np.sort(np.concatenate([rng.uniform(99.5, 100.0, 50000), rng.uniform(0.0, 100.0, 100)]))
    uniform = np.sort(rng.uniform(0.0, 125.0, 4000))
    for tabs in ([cluster], [uniform], [cluster, uniform], []):
        for n in (1, 2, 3, 8, 16):
            r = cs.balanced_ranges(nu, tabs, n)
            assert len(r) == n and r[0][0] == 0 and r[-1][1] == len(nu)
            assert all(0 <= a < b <= len(nu) for a, b in r) and all(r[i][1] == r[i + 1][0] for i in range(n - 1))
            assert all(a % 64 == 0 for a, _ in r)                     # 100037 >= 64 * 4 * 16: every interior edge on a tile boundary
    # equal cost: a uniform table on a uniform grid gives equal ranges (to a tile)
    r = cs.balanced_ranges(np.linspace(1.0, 2500.0, 64000), [np.sort(rng.uniform(0.0, 2525.0, 50000))], 8)
    sizes = np.array([b - a for a, b in r])
    assert sizes.max() - sizes.min() <= 0.2 * sizes.mean()             # (cost grows with nu -- Doppler widths: later ranges are a little shorter)
    assert np.all(np.diff(sizes) <= 64)
    # short grids: one point per range is the limit, more parts than points an error
    assert cs.balanced_ranges(np.linspace(1.0, 2.0, 10), [], 10) == [(i, i + 1) for i in range(10)]
    with pytest.raises(cs.ClearSkyHIPError):
        cs.balanced_ranges(np.linspace(1.0, 2.0, 10), [], 11)


def test_rebalance_ranges_from_measured_times():
    """cs_rebalance_ranges: a partition re-cut from what its ranges were measured to take (the column's own behaviour instead of the cost
    model's constants, fitted to BASELINE configs[2]).  A synthetic 'true' cost that the model misses -- twice as expensive in the upper
    half of the grid, plus a fixed share per range -- is balanced to a few per cent in two passes; the result is always a partition."""
    import clearsky_jl_amd as cs
    rng = np.random.default_rng(7)
    nu = np.linspace(1.0, 2500.0, 40000)
    tabs = [np.sort(rng.uniform(0.0, 2525.0, 30000)), np.sort(rng.uniform(500.0, 900.0, 8000))]
    n = 8
    truth = np.where(nu > 1250.0, 2.0, 1.0) * (1.0 + 0.5 * np.sin(nu / 200.0) ** 2)     # cost per point nobody told the model about
    csum = np.concatenate([[0.0], np.cumsum(truth)])
    fixed = 0.15 * csum[-1] / n

    def measure(ranges):
        return [fixed + csum[b] - csum[a] for a, b in ranges]

    def check_partition(ranges):
        assert ranges[0][0] == 0 and ranges[-1][1] == len(nu)
        assert all(a < b for a, b in ranges) and all(ranges[i][1] == ranges[i + 1][0] for i in range(n - 1))
        assert all(a % 64 == 0 for a, _ in ranges)

    r0 = cs.balanced_ranges(nu, tabs, n)
    t0 = measure(r0)
    r1 = cs.rebalance_ranges(nu, tabs, r0, t0, fixed_time=0.3 * min(t0))
    check_partition(r1)
    t1 = measure(r1)
    r2 = cs.rebalance_ranges(nu, tabs, r1, t1, fixed_time=0.3 * min(t1))
    check_partition(r2)
    t2 = measure(r2)
    spread = lambda t: (max(t) - min(t)) / (sum(t) / len(t))
    assert spread(t0) > 0.3                      # the model alone is far off on this column
    assert spread(t1) < 0.5 * spread(t0) and spread(t2) < 0.05
    assert max(t2) < 0.85 * max(t0)              # what an N-GPU step waits for: its slowest range
    # equal measured times on the model's own partition: nothing to correct
    cum_same = cs.rebalance_ranges(nu, tabs, r0, [1.0] * n, fixed_time=0.0)
    assert cum_same == r0
    # refusals: not a partition, non-positive times
    bad = list(r0)
    bad[3] = (bad[3][0] + 64, bad[3][1])
    with pytest.raises(cs.ClearSkyHIPError):
        cs.rebalance_ranges(nu, tabs, bad, t0)
    with pytest.raises(cs.ClearSkyHIPError):
        cs.rebalance_ranges(nu, tabs, r0, [0.0] + t0[1:])
