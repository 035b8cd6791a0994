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
