"""The bench line the driver parses: flags, metric string, and (on a GPU) every key of the contract."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_metric_is_baseline_jsons_and_defaults_are_the_contracts(monkeypatch):
    m = _bench()
    assert m.baseline_metric() == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = m.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 1 and not a.with_upload and a.queue == 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    a = m.parse()
    assert (a.gpus, a.steps, a.warmup) == (4, 7, 3)


@pytest.mark.gpu
def test_bench_line_carries_every_key_of_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--scenes", "2", "--steps", "2",
                          "--warmup", "1", "--cpu-frames", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["value"] > 0 and abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
