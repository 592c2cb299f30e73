"""-m gpu: bench.py prints ONE JSON line with the keys the driver reads (task contract), on a small workload so that it runs in
seconds; the default workload's line is produced by the driver itself at round end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3",
                        "--workload", "il_onelayer"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["unit"] == "updates/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-6)
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0.0 < r["frac"] < 1.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
