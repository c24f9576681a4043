"""The relevancy storage format either side of the hot path, on HIP kernels (csrc/relio.hip):

    pack_relevancy    what `generate_relevancy.generate_saliency_helper` does to the output of `get_clip_saliency` before it is written
                      (generate_relevancy.py:95-118): nearest-exact resize to the storage dims, "mean" row, normalised text features
    unpack_relevancy  what `dataset.py` does after reading it back (dataset.py:821-871, :1053): pick the label rows, subtract the "mean"
                      map, bilinear up to the image size, x 50

The HDF5 container (gzip chunks, region references, file locks) is storage and is not reproduced here; these functions take and return
device tensors so a caller can put any container around them.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib


def pack_relevancy(maps: torch.Tensor, text_features: torch.Tensor, storage_dims: Tuple[int, int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """maps fp32 [L, H, W], text_features fp32 [L, E] -> (stored fp32 [L + 1, h, w], features fp32 [L + 1, E]); last row = "mean"."""
    dev = _lib.require_gpu()
    maps = maps.to(dev, torch.float32).contiguous()
    feats = text_features.to(dev, torch.float32).contiguous()
    L, H, W = (int(v) for v in maps.shape)
    assert feats.shape[0] == L and L > 0
    h, w = int(storage_dims[0]), int(storage_dims[1])
    stored = torch.empty(L + 1, h, w, dtype=torch.float32, device=dev)
    out_f = torch.empty(L + 1, feats.shape[1], dtype=torch.float32, device=dev)
    st = _lib.stream()
    _lib.call("semabs_relevancy_pack", _lib.ptr(maps), _lib.ptr(stored), L, H, W, h, w, st)
    _lib.call("semabs_text_pack", _lib.ptr(feats), _lib.ptr(out_f), L, int(feats.shape[1]), st)
    return stored, out_f


def unpack_relevancy(stored: torch.Tensor, image_shape: Tuple[int, int], rows: Optional[Sequence[int]] = None, mean_index: Optional[int] = None,
                     scale: float = 1.0) -> torch.Tensor:
    """stored fp32 [R, h, w]; rows = indices of the wanted label rows (default: all); mean_index = row of the "mean" map to subtract
    (`subtract_mean_relevancy`), or None; -> fp32 [P, H, W] = scale * bilinear(stored[rows] - stored[mean_index])."""
    dev = _lib.require_gpu()
    stored = stored.to(dev, torch.float32).contiguous()
    R, h, w = (int(v) for v in stored.shape)
    idx = torch.arange(R, dtype=torch.int64, device=dev) if rows is None else torch.as_tensor(list(rows), dtype=torch.int64, device=dev)
    P = int(idx.numel())
    H, W = int(image_shape[0]), int(image_shape[1])
    out = torch.empty(P, H, W, dtype=torch.float32, device=dev)
    mean_map = None if mean_index is None else stored[int(mean_index)]
    _lib.call("semabs_relevancy_unpack", _lib.ptr(stored), _lib.ptr(idx), _lib.ptr(mean_map), _lib.ptr(out), P, h, w, H, W, float(scale),
              _lib.stream())
    return out
