"""ORACLE (test infrastructure, not product code) - CPU restatement of the relevancy storage format either side of the path:
`generate_saliency_helper`'s post-processing (generate_relevancy.py:95-118) and the loader recipe (dataset.py:821-871, x 50 at :1053).
The arithmetic is torch's own (`interpolate` nearest-exact / bilinear, mean, norm) - the same calls the reference makes.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.  Pinned by tests/golden/g19_relevancy_storage.npz (the reference's two functions executed from source)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def pack_relevancy(maps: torch.Tensor, text_features: torch.Tensor, storage_dims):
    """maps [L, H, W], text_features [L, E] -> (stored [L + 1, h, w], features [L + 1, E])   generate_relevancy.py:95-118"""
    s = F.interpolate(maps[:, None, :, :], size=tuple(int(d) for d in storage_dims), mode="nearest-exact")[:, 0]
    s = torch.cat([s, s.mean(dim=0, keepdim=True)], dim=0)
    f = torch.cat([text_features, text_features.mean(dim=0, keepdim=True)], dim=0)
    f = f / f.norm(dim=-1, keepdim=True)
    return s, f


def unpack_relevancy(stored: torch.Tensor, image_shape, rows=None, mean_index=None, scale: float = 1.0):
    """stored [R, h, w] -> [P, H, W]   dataset.py:821-832 (row pick, - mean), :866-871 (bilinear, align_corners=False), :1053 (x 50)"""
    p = stored if rows is None else stored[list(rows)]
    p = p.float().clone()
    if mean_index is not None:
        p -= stored[mean_index].float().squeeze()
    p = F.interpolate(p[:, None, :, :], size=tuple(int(d) for d in image_shape), mode="bilinear", align_corners=False)[:, 0]
    return p * scale
