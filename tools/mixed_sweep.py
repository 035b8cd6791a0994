"""Tolerance sweep of the fp32 mixed-precision variant (BASELINE configs[4]): sigma / tau / OLR differences from the fp64 path
and step time as a function of far_s.  Prints a markdown table (copied into DESIGN.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clearsky_jl_amd as cs
import workloads as W

cfg = W.config(sys.argv[1] if len(sys.argv) > 1 else "C3")
ctx = cs.Context(0)
def run(mode, far_s):
    ctx.set_precision(mode, far_s)
    col = cs.Column(cfg["P"], cfg["g"], cfg["T"], cfg["mu"], cfg["fS"], cfg["fa"], *cfg["absorbers"], core=cfg["core"],
                    theta_s=cfg["theta_s"], want_tau=True, want_M=False, ctx=ctx)
    col.run(); col.sync()
    t0 = time.perf_counter()
    for _ in range(10): col.run()
    col.sync()
    ms = (time.perf_counter() - t0) * 100
    tau = np.zeros((col.nl, col.nnu), order="F")
    Fup, Fdn = col.fetch(tau)
    return col.sigma_nodes(), tau, Fup, Fdn, ms
s0, t0_, Fu0, Fd0, ms0 = run("fp64", 1e6)
print("| far_s | pairs in fp32 | ms/step | max rel diff sigma | max rel diff tau | OLR diff [W/m2] | max |dF| [W/m2] |")
print("|---|---|---|---|---|---|---|")
print(f"| fp64 | 0 | {ms0:.2f} | 0 | 0 | 0 | 0 |")
for far_s in (1e6, 1e7, 1e8, 1e9):
    s1, t1, Fu1, Fd1, ms1 = run("mixed", far_s)
    m = s0 > 0
    print(f"| {far_s:.0e} | - | {ms1:.2f} | {np.max(np.abs(s1[m]/s0[m]-1)):.2e} | {np.max(np.abs(t1/t0_-1)):.2e} | {Fu1[0]-Fu0[0]:+.2e} | "
          f"{max(np.max(np.abs(Fu1-Fu0)), np.max(np.abs(Fd1-Fd0))):.2e} |")
