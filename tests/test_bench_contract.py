"""bench.py's contract with the driver: one JSON line with the agreed keys (GPU), and a loud failure without a HIP device (CPU)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0.05 < rf["frac"] < 1.0 and rf["launches"] > 0
    # nothing in the line may claim more than the hardware has: no class above its roof, no stream above the HBM peak (round 3 shipped a LayerNorm row at
    # 1.02 of its roof / 8.18 TB/s because a flag word was read without its mask)
    for cls, row in d.get("stages", {}).get("kernel_classes", {}).items():
        assert row.get("frac_of_roof") is None or row["frac_of_roof"] <= 1.0, (cls, row)
        assert row.get("tb_per_s") is None or row["tb_per_s"] <= 8.0, (cls, row)
        assert row.get("tflops") is None or row["tflops"] <= 2500.0, (cls, row)
    for name, row in rf.get("per_shape", {}).items():
        assert row["frac_of_roof"] <= 1.0, (name, row)
    # imported counter / probe figures must come from THIS round's profiles and say that they are imported
    assert "profiles/r04_" in (rf.get("traffic_source") or "profiles/r04_"), rf.get("traffic_source")
    assert rf["mfma_probe"] is None or "imported" in rf["mfma_probe"]["source"]
    assert d.get("text_tower_in_timed_region") is True
