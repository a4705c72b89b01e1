"""GPU: the bench.py contract the driver relies on - one JSON line on stdout with the metric BASELINE.json names, the roofline
object of the dominant kernel and the CPU baseline - on the small workload (c0) so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline(gpu_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c0", "--steps", "3", "--warmup", "1", "--train-steps", "2",
                          "--no-glow-variant"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "hypotheses/s" and "hypotheses" in str(base.get("metric", "hypotheses")) and d["value"] > 0
    assert abs(d["value"] - d["config"]["images_per_gpu"] * d["config"]["hypotheses_per_image"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and "kernel" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"] and c["sample"]
    assert d["train_step"]["img_per_s"] > 0
    # >= 3 timed windows, the median one reported (box noise: VERDICT r4 weak #9)
    assert len(d["windows_ms"]) == 3 and abs(sorted(d["windows_ms"])[1] - d["ms_per_step"]) < 2e-3
    assert len(d["train_step"]["windows_ms"]) == 3
    # the reference's real loop body: train step + sample(N=200) + MHEntLoss metrics (hand/CrossModalHand.py:349-361)
    it = d["train_step"]["iteration_with_metrics"]
    assert "error" not in it, it
    assert it["test_samples"] == 200 and it["ms_per_step"] >= d["train_step"]["ms_per_step"] * 0.9 and it["img_per_s"] > 0
