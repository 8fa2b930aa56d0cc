"""bench.py's line as the driver reads it: every contract field present, no swallowed error (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line_carries_roofline_cpu_baseline_and_parity():
    """One short run of the default configuration (W32 256x192, batch 64, flip test, split fp16; the other configs
    switched off to keep it to a minute).  The blocks computed after the timed region are wrapped in try / except so that
    nothing can keep the headline from being printed -- which once hid a failing parity block: none of them may hold an
    `error`, and the parity block must gate the mode that is reported as `value`."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-other-configs"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["value"] > 0 and abs(line["value"] - 64 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    assert "workload" in line["config"] and line["dtype"].startswith("f16x2")
    roof = line["roofline"]
    assert "error" not in roof and roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert "error" not in cb and cb["value"] > 0 and cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    par = line["parity_vs_cpu_oracle"]
    assert "error" not in par
    assert par["f16x2"]["heatmap_max_abs_err"] < 1e-3 and par["f16x2"]["argmax_equal_rate"] == 1.0
    assert par["f32"]["heatmap_max_abs_err"] < 1e-3 and par["f32"]["argmax_equal_rate"] == 1.0
    for mode in line.get("other_modes", {}).values():
        assert "error" not in mode
