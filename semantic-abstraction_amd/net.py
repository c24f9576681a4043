"""Drop-in for the on-path classes of the reference's `net.py`: `VirtualGrid` (net.py:24-201) and `SemAbs3D`
(net.py:319-439), computed by HIP kernels.  Baselines (SemanticAware*, ClipSpatialVOOL) are out of scope.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from . import _lib


class VirtualGrid:
    def __init__(self, scene_bounds, grid_shape: Tuple[int, int, int] = (32, 32, 32), batch_size: int = 8,
                 device=None, int_dtype=torch.int64, float_dtype=torch.float32, reduce_method: str = "mean"):
        self.lower_corner = tuple(scene_bounds[0])
        self.upper_corner = tuple(scene_bounds[1])
        self.grid_shape = tuple(int(s) for s in grid_shape)
        self.batch_size = int(batch_size)
        self.device = device
        self.int_dtype = int_dtype
        self.float_dtype = float_dtype
        self.reduce_method = reduce_method
        # fp32 constants formed with the reference's op order (net.py:91-95): offsets = -lc, scales = (S-1)/(uc-lc)
        lc = np.asarray(self.lower_corner, np.float64).astype(np.float32)
        uc = np.asarray(self.upper_corner, np.float64).astype(np.float32)
        idx_scale = np.asarray(self.grid_shape, np.float32) - np.float32(1)
        self.offsets = (-lc).astype(np.float32)
        self.scales = (idx_scale / (uc - lc)).astype(np.float32)

    @property
    def num_grids(self):
        return int(np.prod((self.batch_size,) + self.grid_shape))

    def flat_idxs(self, points: torch.Tensor) -> torch.Tensor:
        """points fp32 [..., 3] on the GPU -> int64 [...] flat voxel index (get_points_grid_idxs + flatten_idxs)."""
        _lib.require_gpu()
        pts = points.contiguous().view(-1, 3)
        flat = torch.empty(pts.shape[0], dtype=torch.int64, device=pts.device)
        _lib.call("semabs_voxel_index", _lib.ptr(pts), pts.shape[0], _lib.farr(self.offsets), _lib.farr(self.scales),
                  _lib.iarr(self.grid_shape), _lib.ptr(flat), None, _lib.stream())
        return flat.view(points.shape[:-1])

    def get_points_grid_idxs(self, points: torch.Tensor, cast_to_int=True, batch_idx=None):
        assert cast_to_int and batch_idx is None, "only the integer, un-batched form is on the path"
        _lib.require_gpu()
        pts = points.contiguous().view(-1, 3)
        idx3 = torch.empty(pts.shape[0], 3, dtype=torch.int32, device=pts.device)
        _lib.call("semabs_voxel_index", _lib.ptr(pts), pts.shape[0], _lib.farr(self.offsets), _lib.farr(self.scales),
                  _lib.iarr(self.grid_shape), None, _lib.ptr(idx3), _lib.stream())
        return idx3.to(self.int_dtype).view(*points.shape[:-1], 3)

    def flatten_idxs(self, idxs: torch.Tensor, keepdim=False):
        S0, S1, S2 = self.grid_shape
        assert idxs.shape[-1] == 3
        flat = idxs[..., 0] * (S1 * S2) + idxs[..., 1] * S2 + idxs[..., 2]
        return flat.unsqueeze(-1) if keepdim else flat
