"""The bench line: one JSON object on stdout carrying the throughput fields, the roofline object of the dominant
kernel (HIP-event timing) and, at one GPU, the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--batch", "256",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["warmup"] == 3
    assert out["unit"] == "samples/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert "workload" in out["config"] and out["config"]["global_batch"] == 256
    # value and ms_per_step describe the same timed region
    assert abs(out["value"] - 256 / (out["ms_per_step"] * 1e-3)) <= 2e-3 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and rf["avg_launch_us"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-3
    assert abs(rf["achieved"] - rf["flops_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e12) <= 1e-2 * rf["achieved"]
    assert out["final_loss"] > 0 and out["final_loss"] == out["final_loss"]
