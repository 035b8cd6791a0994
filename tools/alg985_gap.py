#!/usr/bin/env python3
"""How far is the exact-Faddeeva path (oracle, kernels) from what the Julia reference would print?

The reference evaluates fvoigt with Faddeyeva985.faddeyeva (ACM TOMS Algorithm 985, line_shapes.jl:375), whose source and version
are not available here.  The oracle carries a restatement of the published algorithm ("alg985" back-end, labelled unverifiable);
this script (CPU only) reports, for BASELINE configs[1] (C2) and configs[2] (C3):
  * the algorithm's own error against scipy's wofz (self-consistency with the paper's "< 4e-5"),
  * max |sigma_985 / sigma_exact - 1| over all (nu, node) with sigma > 0, the same for tau,
  * OLR_985 - OLR_exact [W/m^2] and the largest band-flux difference.
Usage:  python tools/alg985_gap.py [--c3-stride 4]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.special import wofz  # noqa: E402

import clearsky_jl_amd as cs  # noqa: E402  (host-side closures only: no GPU call is made)
import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3-stride", type=int, default=4, help="evaluate every n-th wavenumber of the C3 grid (both back-ends on the same sub-grid)")
    args = ap.parse_args()
    out = {}
    rng = np.random.default_rng(0)
    n = 400000
    x = np.concatenate([rng.uniform(0, 8, n), 10 ** rng.uniform(-3, 4, n)])
    y = np.concatenate([10 ** rng.uniform(-8, 1, n), 10 ** rng.uniform(-6, 3, n)])
    ex = wofz(x + 1j * y).real
    with O.faddeeva_backend("alg985"):
        a = O.faddeeva(x, y)
    with O.faddeeva_backend("exact"):
        b = O.faddeeva(x, y)
    out["alg985_vs_wofz_max_rel"] = float(np.max(np.abs(a / ex - 1)))
    out["exact_vs_wofz_max_rel"] = float(np.max(np.abs(b / ex - 1)))
    for name, stride in (("C2", 1), ("C3", args.c3_stride)):
        cfg = W.config(name)
        nu = np.ascontiguousarray(cfg["nu"][::stride])
        P = cfg["P"]
        fT, fmu = cs.formprofile(P, cfg["T"]), cs.formprofile(P, cfg["mu"])
        Tn, mun = cs.lobattoevaluations(P, fT, fmu, 2)
        Tk, Pk = cs.nodevalues(Tn, 2), cs.nodepressures(P, 2)
        gases = [g for g in cfg["absorbers"] if isinstance(g, cs.DirectGas)]
        conc = np.array([[g.fC(Tk[k], Pk[k]) for k in range(len(Pk))] for g in gases])
        kw = dict(theta_s=cfg["theta_s"], nstream=5, want_sigma=True)
        res = {}
        for be in ("exact", "alg985"):
            with O.faddeeva_backend(be):
                res[be] = O.fluxes_discretized(nu, P, cfg["g"], 2, Tn, mun, np.array([fT(p) for p in P]), [g.sl for g in gases],
                                               ["voigt"] * len(gases), [25.0] * len(gases), conc, **kw)
        e, r = res["exact"], res["alg985"]
        m = e["sigma"] > 0
        w = cs.trapz_weights(cfg["nu"])[::stride] * stride          # band integrals of the sub-grid (same for both back-ends)
        out[name] = dict(points=len(nu), stride=stride,
                         sigma_max_rel=float(np.max(np.abs(r["sigma"][m] / e["sigma"][m] - 1))),
                         sigma_median_rel=float(np.median(np.abs(r["sigma"][m] / e["sigma"][m] - 1))),
                         tau_max_rel=float(np.max(np.abs(r["tau"] / e["tau"] - 1))),
                         olr_exact=float(e["Fup"][0]), olr_985_minus_exact=float(r["Fup"][0] - e["Fup"][0]),
                         flux_max_abs_diff=float(max(np.max(np.abs(r["Fup"] - e["Fup"])), np.max(np.abs(r["Fdn"] - e["Fdn"])))),
                         olr_rel=float((r["Fup"][0] - e["Fup"][0]) / e["Fup"][0]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
