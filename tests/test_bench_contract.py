"""The driver's bench.py contract: one JSON line with the agreed keys (run on a small configuration)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "3", "--warmup", "1",
                          "--cpu-stride", "16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "spectral-points/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["value"] == pytest.approx(10_000 * 40 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma", "valu_issue") and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0
    assert abs(d["olr_wm2"] - c["olr_sample"]) < 5.0          # the CPU leg ran a 1/16 sample of the same column
