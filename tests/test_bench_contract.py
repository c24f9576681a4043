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
    src = rf.get("traffic_source")
    assert src is None or (("profiles/r06_" in src or "profiles/r05_" in src) and "imported" in src), src
    assert "mfma_probe" not in rf                              # round-4 probe figures are no longer carried forward
    assert d.get("text_tower_in_timed_region") is True
    # clock and power under load, sampled in THIS run: the keys are always there; on a box with any SMI source they carry numbers
    for k in ("sclk_mhz_under_load", "power_w", "power_cap_w", "smi_source", "smi_samples"):
        assert k in rf, k
    if rf["smi_source"] is not None:
        assert rf["smi_samples"] > 0 and (rf["sclk_mhz_under_load"] or rf["power_w"])
        assert rf["sclk_mhz_under_load"] is None or 100 < rf["sclk_mhz_under_load"] < 3000
        assert rf["power_w"] is None or 50 < rf["power_w"] < 2000
        if rf["sclk_mhz_under_load"]:                       # the clock-adjusted context figure: same rate, peak scaled to the sampled clock; `frac` itself stays against 2.4 GHz
            assert abs(rf["frac_at_measured_sclk"] * rf["peak_at_measured_sclk"] - rf["achieved"]) < 1e-6 * rf["achieved"]
            assert abs(rf["peak_at_measured_sclk"] - rf["peak"] * rf["sclk_mhz_under_load"] / 2400.0) < 1e-6 * rf["peak"]
            assert abs(rf["frac"] * rf["peak"] - rf["achieved"]) < 1e-6 * rf["achieved"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra,metric_part", [(["--mode", "latency"], "LATENCY"), (["--workload", "train"], "VOOL optimisation steps")])
def test_bench_multi_rank_modes_over_gloo_on_one_gpu(extra, metric_part):
    """The two multi-GPU modes SURVEY 8(e) names besides scene sharding, as bench lines: two ranks sharing this box's one GPU over
    gloo (RCCL admits one rank per device).  Latency mode: one scene tile- and label-sharded over the ranks, every rank ends with identical maps and
    labels (checksums in the line) and the all-gather payloads are stated; train workload: the data-parallel VOOL step, identical parameters on
    both ranks after the flat gradient all-reduce, whose bytes are stated."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-parity", "--no-stages"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and metric_part in d["metric"] and d["value"] > 0 and d["config"]["backend"] == "gloo"
    c = d["collectives"]
    # every N > 1 line says what each rank put on the wire and how long its host spent issuing it
    assert len(c["per_rank"]) == 2 and sorted(r["rank"] for r in c["per_rank"]) == [0, 1]
    assert all(sum(k["bytes"] for k in r["collectives"].values()) > 0 for r in c["per_rank"])
    # ... and how long the COMPUTE stream spent in / waiting for communication (device-side events)
    assert c["exposed_ms_per_step"] is not None and c["exposed_ms_per_step"] >= 0.0
    assert all("device_ms" in k for r in c["per_rank"] for k in r["collectives"].values())
    if "--mode" in extra:
        assert d["scaling"] == "strong" and c["identical_across_ranks"] is True and len(c["maps_checksums_by_rank"]) == 2
        assert c["tile_relevance_allgather_bytes_per_rank_per_scene"] == 2 * 16 * 612 * 14 * 14 * 4          # 1224 tiles / 2 ranks, 16 labels, 14 x 14, two flip passes
        assert c["logits_allgather_bytes_per_rank_per_scene"] == 8 * 128 ** 3 * 4 and c["gather_results_ranks_seen"] == 2
    else:
        assert d["scaling"] == "weak" and c["parameters_identical_across_ranks"] is True and c["allreduce_bytes_per_rank_per_step"] > 100e6 and c["allreduce_ms_alone"] > 0
        # the gradient exchange runs in >= 4 buckets announced by the backward pass; together they are the whole flat buffer
        assert c["overlapped_with_backward"] is True and len(c["buckets_bytes"]) >= 4 and sum(c["buckets_bytes"]) == c["allreduce_bytes_per_rank_per_step"]
        assert all(r["collectives"]["all_reduce_bucket"]["calls"] == 4 * r["steps"] for r in c["per_rank"])
