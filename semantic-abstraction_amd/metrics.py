"""Drop-in for the evaluation metrics that follow the path: `utils.voxelize_points` (utils.py:617-665),
`utils.prediction_analysis` and `utils.iou` (utils.py:329-380), on HIP kernels (csrc/evalm.hip).  The device does the point / voxel passes
with integer atomics and returns exact counts; the ratios are then formed on the host with the reference's own operand types."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from .net import VirtualGrid


def _u8(t: torch.Tensor, dev) -> torch.Tensor:
    return (t.to(dev) != 0).to(torch.uint8).contiguous()


def voxelize_points(prediction, label, xyz_pts, voxel_shape: Tuple[int, int, int], scene_bounds, ignore_pts, device="cuda") -> Dict[str, torch.Tensor]:
    """prediction / label / ignore_pts [B, P, N] (bool-like), xyz_pts [B, P, N, 3] -> dict(prediction, label, ignore) of [B, P, S^3];
    `prediction` and `ignore` bool, `label` float (0 / 1), like the reference."""
    dev = _lib.require_gpu()
    B, P, N = (int(v) for v in prediction.shape)
    vg = VirtualGrid(scene_bounds=np.asarray(scene_bounds), grid_shape=tuple(voxel_shape), batch_size=B * P, reduce_method="max")
    nvox = int(np.prod(voxel_shape))
    flat = vg.flat_idxs(xyz_pts.to(dev, torch.float32).reshape(B * P * N, 3)).contiguous()
    pr, lb, ig = _u8(prediction.reshape(B * P, N), dev), _u8(label.reshape(B * P, N), dev), _u8(ignore_pts.reshape(B * P, N), dev)
    scratch = torch.empty(B * P, nvox, 3, dtype=torch.int32, device=dev)
    out = [torch.empty(B * P, nvox, dtype=torch.uint8, device=dev) for _ in range(3)]
    _lib.call("semabs_voxelize_eval", _lib.ptr(flat), _lib.ptr(pr), _lib.ptr(lb), _lib.ptr(ig), _lib.ptr(scratch), _lib.ptr(out[0]), _lib.ptr(out[1]),
              _lib.ptr(out[2]), B * P, N, nvox, _lib.stream())
    return {"prediction": out[0].view(B, P, nvox).bool(), "label": out[1].view(B, P, nvox).float(), "ignore": out[2].view(B, P, nvox).bool()}


def prediction_counts(prediction, label, ignore) -> np.ndarray:
    """[B, P, M] bool-like -> int64 [B, P, 5]: valid, positive labels, positive predictions, true positives, union (ignored elements excluded)."""
    dev = _lib.require_gpu()
    B, P, M = (int(v) for v in prediction.shape)
    pr, lb, ig = _u8(prediction.reshape(B * P, M), dev), _u8(label.reshape(B * P, M), dev), _u8(ignore.reshape(B * P, M), dev)
    counts = torch.empty(B * P, 6, dtype=torch.int64, device=dev)
    _lib.call("semabs_prediction_counts", _lib.ptr(pr), _lib.ptr(lb), _lib.ptr(ig), _lib.ptr(counts), B * P, M, _lib.stream())
    return counts.cpu().numpy().reshape(B, P, 6)[..., :5]


def prediction_analysis(prediction, label, ignore) -> Dict[str, List[float]]:
    """Same keys, order (scene-major) and values as the reference: python floats, NaN where the reference produces NaN."""
    c = prediction_counts(prediction, label, ignore)
    stats = {"precision": [], "recall": [], "false_negative": [], "false_positive": [], "iou": []}
    f32 = np.float32
    for b in range(c.shape[0]):
        for p in range(c.shape[1]):
            n, pl, pp, tp, un = (int(v) for v in c[b, p])
            with np.errstate(divide="ignore", invalid="ignore"):
                stats["iou"].append(float(f32(tp) / f32(un)))                      # fp32 division of two .float() sums (utils.py:335-337)
                stats["precision"].append(float(f32(tp)) / float(f32(pp)) if pp != 0 else float("nan"))
                stats["recall"].append(float(f32(tp)) / float(f32(pl)) if pl != 0 else float("nan"))
                stats["false_negative"].append(float(f32(pl - tp) / f32(n)))      # .float().mean(): exact fp32 sum / count
                stats["false_positive"].append(float(f32(pp - tp) / f32(n)))
    return stats


def iou(prediction, label) -> torch.Tensor:
    """[..., N] bool-like -> fp32 [...] intersection / union (utils.py:329-337)."""
    shape = tuple(prediction.shape[:-1])
    n = int(prediction.shape[-1])
    c = prediction_counts(prediction.reshape(1, -1, n), label.reshape(1, -1, n), torch.zeros(1, int(np.prod(shape)) if shape else 1, n, dtype=torch.uint8))
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (c[0, :, 3].astype(np.float32) / c[0, :, 4].astype(np.float32)).reshape(shape)
    return torch.from_numpy(np.asarray(r, np.float32))
