"""bench.py's one-JSON-line contract on a GPU box, at sizes that take seconds: the keys the driver and the judge read
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, the `roofline` object with frac <= 1 and the `cpu_baseline` object) for the headline path, a family
line and the side workloads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "iterations/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) <= 1e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert 0.0 < d["roofline_iteration"]["frac"] <= 1.0


@pytest.mark.timeout(900)
def test_headline_line_small():
    d = _run("--size", "400000", "--steps", "30", "--warmup", "5", "--cpu-states", "3", "--no-extras")
    _check_common(d, 30, 5)
    assert d["dtype"] == "f64" and d["roofline"]["kernel"].startswith("bz::k_fused_compact<XR=2")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["reference_dataflow"]["speedup"] > 1.0          # (a ratio beside the roofline, never inside it)


@pytest.mark.timeout(900)
def test_family_line_small():
    d = _run("--family", "diag-nonneg-eitheror", "--size", "400000", "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
             "--no-extras")
    _check_common(d, 30, 5)
    assert "FAM=" in d["roofline"]["kernel"] and d["config"]["family"] == "diag-nonneg-eitheror"


@pytest.mark.timeout(900)
def test_als_line_small():
    d = _run("--workload", "als", "--size", "300000", "--steps", "20", "--warmup", "4")
    _check_common(d, 20, 4)
    assert d["config"]["workload"].startswith("als") and d["cpu_baseline"]["kind"] == "port"
