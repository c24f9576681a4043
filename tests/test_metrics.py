"""Evaluation metrics after the path: oracle vs the golden of the reference's own functions (CPU); HIP kernels vs the golden
(GPU) - integer work, so everything is compared exactly (NaN == NaN)."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import metrics as om
from semabs_amd.synth import SCENE_BOUNDS

KEYS = ("precision", "recall", "false_negative", "false_positive", "iou")


def _same(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _inputs(g):
    return (torch.from_numpy(g["pred"]), torch.from_numpy(g["label"]), torch.from_numpy(g["ignore"]), torch.from_numpy(g["xyz"]),
            int(g["meta"][3]))


def test_oracle_metrics_match_reference(golden):
    g = golden("g15_metrics")
    pred, label, ignore, xyz, S = _inputs(g)
    vox = om.voxelize_points(pred, label, xyz, (S, S, S), SCENE_BOUNDS, ignore)
    assert np.array_equal(vox["prediction"].numpy(), g["vox_prediction"])
    assert np.array_equal(vox["label"].numpy(), g["vox_label"])
    assert np.array_equal(vox["ignore"].numpy(), g["vox_ignore"])
    pts, vst = om.prediction_analysis(pred, label, ignore), om.prediction_analysis(**vox)
    for k in KEYS:
        assert _same(pts[k], g["point_" + k]), k
        assert _same(vst[k], g["voxel_" + k]), k
    assert _same(om.iou(pred, label).numpy(), g["iou_rows"])
    assert np.isnan(g["point_precision"]).any() and np.isnan(g["point_iou"]).any()          # the fixture exercises the NaN branches


@pytest.mark.gpu
def test_hip_metrics_match_reference_exactly(golden):
    from semabs_amd import metrics as hm
    g = golden("g15_metrics")
    pred, label, ignore, xyz, S = _inputs(g)
    vox = hm.voxelize_points(pred, label, xyz, (S, S, S), SCENE_BOUNDS, ignore)
    assert np.array_equal(vox["prediction"].cpu().numpy(), g["vox_prediction"])
    assert np.array_equal(vox["label"].cpu().numpy(), g["vox_label"])
    assert np.array_equal(vox["ignore"].cpu().numpy(), g["vox_ignore"])
    pts, vst = hm.prediction_analysis(pred, label, ignore), hm.prediction_analysis(**vox)
    for k in KEYS:
        assert _same(pts[k], g["point_" + k]), k
        assert _same(vst[k], g["voxel_" + k]), k
    assert _same(hm.iou(pred, label).numpy(), g["iou_rows"])


@pytest.mark.gpu
def test_hip_metrics_full_size_properties():
    """400 000 points x 4 descriptions into 64^3: counts are consistent (tp <= min(labels, predictions), union = labels + predictions - tp),
    a perfect prediction has IoU 1 / no false rates, and an all-ignored row yields NaN."""
    from semabs_amd import metrics as hm
    rng = np.random.default_rng(3)
    B, P, N = 1, 4, 400000
    label = torch.from_numpy(rng.random((B, P, N)) < 0.2)
    pred = torch.from_numpy(rng.random((B, P, N)) < 0.25)
    ignore = torch.from_numpy(rng.random((B, P, N)) < 0.05)
    ignore[0, 3] = True
    c = hm.prediction_counts(pred, label, ignore)[0]
    assert (c[:, 3] <= np.minimum(c[:, 1], c[:, 2])).all() and (c[:, 4] == c[:, 1] + c[:, 2] - c[:, 3]).all()
    keep = ~ignore[0, 0]
    assert c[0, 0] == int(keep.sum()) and c[0, 3] == int((label[0, 0] & pred[0, 0] & keep).sum())
    st = hm.prediction_analysis(label, label, ignore)
    assert st["iou"][0] == 1.0 and st["false_negative"][0] == 0.0 and st["false_positive"][0] == 0.0 and np.isnan(st["iou"][3])
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = torch.from_numpy((lo + (hi - lo) * rng.random((B, P, N, 3))).astype(np.float32))
    vox = hm.voxelize_points(pred, label, xyz, (64, 64, 64), SCENE_BOUNDS, ignore)
    assert tuple(vox["prediction"].shape) == (B, P, 64 ** 3) and bool(vox["ignore"][0, 3].all())
    assert float(vox["label"].max()) == 1.0 and vox["prediction"].dtype == torch.bool
