"""bench.py's contract on a small frame (the driver runs it at the headline size at the end of every round): ONE JSON line
on stdout with the metric / roofline / cpu_baseline objects, in the default mode, the drop-in mode and with the reference's
evaluation flags.  Runs bench.py as a child process (it owns its own streams and process group state)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gaussians", "30000", "--width", "480", "--height", "272", "--steps", "6", "--warmup", "2", "--forward-frames", "2"]


def _bench(*extra):
    from oracle_farm import FARM
    FARM.drain()            # the GPU boxes admit six processes on the card: the oracle workers exit first
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, *extra], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-1000:]
    return json.loads(lines[0])


def test_default_line_carries_the_contract_fields(gpu_device):
    d = _bench("--cpu-tiles", "40", "--cpu-gaussians", "5000")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["unit"] == "iters/s" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert abs(d["value"] * d["ms_per_step"] - 1000.0) < 1.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"] in ("render_fwd", "render_bwd", "preprocess_fwd", "preprocess_bwd") and r["avg_launch_ms"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "iters/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["mode"].startswith("fused")
    full = d["with_all_seven_allmap_channels"]
    assert full["value"] > 0 and full["steps"] >= 10


def test_dropin_mode_and_evaluation_flags(gpu_device):
    drop = _bench("--no-cpu-baseline", "--mode", "dropin")
    assert drop["config"]["mode"].startswith("dropin") and drop["value"] > 0 and drop["with_all_seven_allmap_channels"] is None
    ev = _bench("--no-cpu-baseline", "--eval-flags")
    assert "lambda_normal 0" in ev["config"]["loss"] and ev["value"] > 0
    # the drop-in loop after the optional edits of INTEGRATION.md section 1 (HIP l1_loss / ssim, FusedAdam, this repository's render())
    swapped = _bench("--no-cpu-baseline", "--mode", "dropin", "--dropin-loss", "hip", "--dropin-adam", "hip", "--dropin-render", "hip")
    mode = swapped["config"]["mode"]
    assert "loss_utils" in mode and "FusedAdam" in mode and "derived maps from one HIP launch" in mode and swapped["value"] > 0
