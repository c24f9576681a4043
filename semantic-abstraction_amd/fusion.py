"""Drop-in for the on-path half of the reference's `fusion.TSDFVolume` (fusion.py:37-209): constructor,
`integrate`, `get_volume`.  The volumes live in HBM; one fused HIP kernel (csrc/geometry.hip:k_tsdf_integrate)
does vox2world -> inverse-pose transform (f64) -> cam2pix (round-half-even, int64) -> frustum test -> depth
lookup -> truncated SDF -> running weighted mean (+ packed colour).  Marching cubes / PLY export are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class TSDFVolume:
    def __init__(self, vol_bnds, voxel_size):
        vol_bnds = np.asarray(vol_bnds)
        assert vol_bnds.shape == (3, 2), f"vol_bnds must be (3, 2): per-axis (min, max) in metres, got {vol_bnds.shape}"
        assert (vol_bnds[:, 0] < vol_bnds[:, 1]).all()
        self._dev = _lib.require_gpu()
        self._vol_bnds = vol_bnds
        self._voxel_size = float(voxel_size)
        self._trunc_margin = 5 * self._voxel_size
        self._color_const = 256 * 256
        self._vol_dim = np.ceil((vol_bnds[:, 1] - vol_bnds[:, 0]) / self._voxel_size).copy(order="C").astype(int)
        # like the reference, the caller's array is adjusted in place (fusion.py:60)
        self._vol_bnds[:, 1] = self._vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_origin = self._vol_bnds[:, 0].copy(order="C").astype(np.float32)
        dims = tuple(int(d) for d in self._vol_dim)
        self._tsdf_vol = _lib.filled(dims, torch.float32, -1.0, self._dev)
        self._weight_vol = _lib.filled(dims, torch.float32, 0, self._dev)
        self._color_vol = _lib.filled(dims, torch.float32, 0, self._dev)
        self.last_pix = None

    @staticmethod
    def params_for(voxel_size, cam_pose, obs_weight=1.0, dev=None) -> torch.Tensor:
        """inv(pose) rows + (voxel size, truncation margin = 5 voxels, observation weight): the 15 doubles semabs_tsdf_integrate reads from the device."""
        T = np.linalg.inv(np.asarray(cam_pose, np.float64))
        vs = float(voxel_size)
        prm = np.concatenate([T[:3, :4].reshape(-1), [vs, 5 * vs, float(obs_weight)]])
        return torch.from_numpy(prm).to(dev if dev is not None else _lib.require_gpu())

    def integrate_params(self, cam_pose, obs_weight=1.0) -> torch.Tensor:
        return TSDFVolume.params_for(self._voxel_size, cam_pose, obs_weight, self._dev)

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.0, keep_pix=False, prm=None):
        im_h, im_w = depth_im.shape[:2]
        dev = self._dev
        depth = depth_im if isinstance(depth_im, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(depth_im, dtype=np.float32))
        depth = depth.to(dev).contiguous()
        color = None
        if color_im is not None:
            color = color_im if isinstance(color_im, torch.Tensor) else torch.from_numpy(
                np.ascontiguousarray(color_im, dtype=np.uint8))
            color = color.to(dev).contiguous()
        if prm is None:
            prm = self.integrate_params(cam_pose, obs_weight)
        K = np.asarray(cam_intr).astype(np.float32)
        pix = torch.empty((self._tsdf_vol.numel(), 2), dtype=torch.int64, device=dev) if keep_pix else None
        _lib.call("semabs_tsdf_integrate", _lib.ptr(color), _lib.ptr(depth), im_h, im_w, _lib.ptr(prm),
                  _lib.farr(self._vol_origin), _lib.farr([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]),
                  _lib.iarr(self._vol_dim), _lib.ptr(self._tsdf_vol), _lib.ptr(self._weight_vol),
                  _lib.ptr(self._color_vol) if color is not None else None, _lib.ptr(pix), _lib.stream())
        self.last_pix = pix

    # numpy views of the device state, under the reference's attribute names
    @property
    def _tsdf_vol_cpu(self):
        return self._tsdf_vol.cpu().numpy()

    @property
    def _weight_vol_cpu(self):
        return self._weight_vol.cpu().numpy()

    @property
    def _color_vol_cpu(self):
        return self._color_vol.cpu().numpy()

    def get_volume(self):
        col = self._color_vol_cpu
        color_vol = np.zeros([3] + list(col.shape), dtype=np.uint8)
        b = np.floor(col / self._color_const)
        g = np.floor((col - b * self._color_const) / 256)
        r = col - b * self._color_const - g * 256
        color_vol[2], color_vol[1], color_vol[0] = b.astype(np.uint8), g.astype(np.uint8), r.astype(np.uint8)
        return self._tsdf_vol_cpu, color_vol
