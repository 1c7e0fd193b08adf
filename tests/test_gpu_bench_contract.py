"""GPU: bench.py prints ONE JSON line that carries the driver's contract (metric, value, unit, n_gpus, steps, warmup,
ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

import scene_feed as sf

pytestmark = pytest.mark.gpu


def test_bench_line_carries_the_contract():
    r = subprocess.run([sys.executable, os.path.join(sf.ROOT, "bench.py"), "--workload", "reflective_world_1080p", "--steps", "3",
                        "--warmup", "1", "--no-extra", "--cpu-sample-stride", "2000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_ms"] <= d["ms_per_step"] * 1.05
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["value"] > 0
    assert d["parity"]["within_tolerance"] is True
