"""ORACLE (test infrastructure, not product code) - CPU restatement of the evaluation metrics that follow the path:
`utils.voxelize_points` (utils.py:617-665, scatter-max of VirtualGrid.scatter_points net.py:185-201), `utils.prediction_analysis` and
`utils.iou` (utils.py:329-380).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
Pinned by tests/golden/g15_metrics.npz (the reference's own functions, run on seeded inputs)."""
from __future__ import annotations

import numpy as np
import torch

from .geometry import flatten_idxs, points_grid_idxs


def _scatter_max(flat: torch.Tensor, feat: torch.Tensor, nvox: int) -> torch.Tensor:
    """flat int64 [B, N], feat fp32 [B, N] -> [B, nvox]; untouched voxels 0 (torch_scatter.scatter(reduce="max") semantics)."""
    out = torch.zeros(flat.shape[0], nvox)
    return out.scatter_reduce(1, flat, feat, reduce="amax", include_self=False)


def voxelize_points(prediction, label, xyz_pts, voxel_shape, scene_bounds, ignore_pts):
    B, P, N = prediction.shape
    nvox = int(np.prod(voxel_shape))
    xyz = xyz_pts.reshape(B * P, N, 3).float()
    flat = torch.from_numpy(flatten_idxs(points_grid_idxs(xyz.numpy(), scene_bounds, voxel_shape), voxel_shape))
    pred = _scatter_max(flat, prediction.reshape(B * P, N).float(), nvox).view(B, P, nvox)
    lab = _scatter_max(flat, (label.reshape(B * P, N).float() - 0.5) * 2, nvox).view(B, P, nvox)
    missing = lab == 0.0
    ign = _scatter_max(flat, ignore_pts.reshape(B * P, N).float(), nvox).view(B, P, nvox).bool()
    return {"prediction": pred > 0, "label": (lab > 0).float(), "ignore": torch.logical_or(ign, missing)}


def iou(prediction, label):
    inter = torch.logical_and(prediction, label).sum(dim=-1).float()
    union = torch.logical_or(prediction, label).sum(dim=-1).float()
    return inter / union


def prediction_analysis(prediction, label, ignore):
    stats = {"precision": [], "recall": [], "false_negative": [], "false_positive": [], "iou": []}
    for b in range(ignore.shape[0]):
        for p in range(ignore.shape[1]):
            mask = ~ignore.bool()[b, p]
            cl = label.bool()[b, p][mask]
            cp = prediction.bool()[b, p][mask]
            pl, pp = cl.float().sum(dim=-1), cp.float().sum(dim=-1)
            tp = torch.logical_and(cl, cp).float().sum(dim=-1)
            stats["iou"].append(iou(cp, cl).item())
            stats["precision"].append(tp.item() / pp.item() if pp.item() != 0 else float("nan"))
            stats["recall"].append(tp.item() / pl.item() if pl.item() != 0 else float("nan"))
            stats["false_negative"].append(torch.logical_and(cl, ~cp).float().mean(dim=-1).item())
            stats["false_positive"].append(torch.logical_and(~cl, cp).float().mean(dim=-1).item())
    return stats
