"""The driver's contract with bench.py, end to end on the GPU: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line
whose metric is BASELINE.json's, whose value is the whole-solve throughput at 1280x720 and which carries the roofline record of the
dominant kernel (priced with counters that belong to the kernel sources in the tree) and the CPU baseline timed on the host."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_line_has_the_contracts_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-side-records"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"].split(",")[0])  # "Mpixels/sec RS depth+pose solve, 1280x720 pair ..."
    assert d["unit"] == "Mpixels/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"].startswith("synthetic")
    assert "1280x720" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 50.0  # north_star's target; the measured value is ~20x that
    assert abs(d["value"] - 1280 * 720 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("fp64-valu", "hbm", "mfma") and r["peak"] > 0 and r["unit"]
    if r["counters_stale"]:  # the kernel sources changed since profiles/counters.json was collected: the line must say so and price nothing
        assert r["counters_stale_files"] and r["frac"] is None and r["achieved"] is None
    else:
        assert 0.3 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["traffic"] is None or r["traffic"] > 0.5 * r["alg_bytes_per_launch"]
    assert r["avg_launch_ms"] > 0
    assert 0.0 < r["hbm"]["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "Mpixels/s" and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"]
