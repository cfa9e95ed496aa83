"""bench.py's result line, as the driver reads it: ONE JSON line on stdout with the contract's keys, `value` = units / wall time of
exactly `steps` steps, a `roofline` whose fraction is achieved / peak and below 1 with the kernel's time inside the step time, and
a `cpu_baseline` timed on the host cores -- run here on a 64^3 mesh so that it takes seconds (the driver runs the default: 256^3)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_line_contract():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--mesh", "64", "--steps", "5", "--warmup", "2"], capture_output=True,
                       text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                      # whatever libraries print goes to stderr
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["unit"] == "cell-updates/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    units = 64 ** 3 * 8 * 5
    assert abs(d["value"] - units / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-9
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    k = d["kernel_ms_per_step"]
    assert k["rates"] <= d["ms_per_step"] and k["column_sweep"] + k["rates"] + k.get("chemistry", 0.0) <= d["ms_per_step"] * 1.001
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["value"] > 20 * cb["value"]                          # (a GPU that does not beat the host cores has fallen back to something)
    assert d["rccl_ranks"] == 0 and d["rccl_library"] is None
