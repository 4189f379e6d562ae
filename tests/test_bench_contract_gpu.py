"""The benchmark's output contract: one JSON line on standard output carrying the metric, its value and the
`roofline` and `cpu_baseline` objects, for the argument form the driver uses (`--gpus N --steps K --warmup W`).
Run at a reduced size so that it takes seconds; the figures themselves are not judged here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--images", "12",
           "--features", "2048", "--ransac-pairs", "16", "--ransac-corr", "2000", "--ransac-hyps", "512", "--cpu-pairs", "1",
           "--no-regimes", "--no-extensions", "--no-pipeline"]
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on standard output"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "pairs/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["ransac"]["value"] > 0 and d["ransac"]["unit"] == "hyps/s"
