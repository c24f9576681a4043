"""Drop-in for the on-path functions of the reference's `point_cloud.py` (numpy in / numpy out), computed
by HIP kernels (csrc/geometry.hip):

  get_pointcloud        point_cloud.py:34-66   (f64 math on device, f64 results returned like the reference)
  filter_pts_bounds     point_cloud.py:24-31
  check_pts_in_frustum  point_cloud.py:88-110

`pointcloud_device` is the fused device-resident form the scene pipeline uses (fp32 points + bounds mask in
one launch, nothing copied to the host).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _params(cam_intr, cam_pose, bounds, dev):
    p = np.zeros(22, np.float64)
    K = np.asarray(cam_intr, np.float64)
    p[0:4] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    if cam_pose is not None:
        p[4:16] = np.asarray(cam_pose, np.float64)[:3, :4].reshape(-1)
    if bounds is not None:
        b = np.asarray(bounds, np.float64)
        p[16:19], p[19:22] = b[0], b[1]
    return torch.from_numpy(p).to(dev)


def pointcloud_params(cam_intr, cam_pose, bounds) -> torch.Tensor:
    return _params(cam_intr, cam_pose, bounds, _lib.require_gpu())


def frustum_params(cam_pose, cam_intr) -> torch.Tensor:
    """inv(pose) rows + (fx, fy, cx, cy) as 16 doubles on the device (the argument block of semabs_frustum_mask)."""
    T = np.linalg.inv(np.asarray(cam_pose, np.float64))
    K = np.asarray(cam_intr, np.float64)
    return torch.from_numpy(np.concatenate([T[:3, :4].reshape(-1), [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]])).to(_lib.require_gpu())


def pointcloud_device(depth: torch.Tensor, cam_intr, cam_pose, bounds=None, prm: torch.Tensor | None = None):
    """depth fp32 [H, W] on the GPU -> (xyz fp32 [H*W, 3], in-bounds mask uint8 [H*W] or None), on the GPU.
    prm: the 22 camera / bounds doubles already on the device (`pointcloud_params`; ScenePipeline.upload puts them there with the frame)."""
    dev = _lib.require_gpu()
    H, W = depth.shape
    xyz = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
    mask = torch.empty(H * W, dtype=torch.uint8, device=dev) if bounds is not None else None
    if prm is None:
        prm = _params(cam_intr, cam_pose, bounds, dev)
    depth = depth.contiguous()
    _lib.call("semabs_pointcloud", _lib.ptr(depth), H, W, _lib.ptr(prm), int(cam_pose is not None),
              _lib.ptr(xyz), _lib.ptr(mask), _lib.stream())
    return xyz, mask


def get_pointcloud(depth_img, color_img, cam_intr, cam_pose=None):
    """-> (f64 [H*W, 3], colour points): the reference's return types (point_cloud.py:51-66); the f64 values are the kernel's own f64
    results, not widened fp32."""
    dev = _lib.require_gpu()
    d = torch.from_numpy(np.ascontiguousarray(depth_img, dtype=np.float32)).to(dev)
    H, W = d.shape
    xyz = torch.empty(H * W, 3, dtype=torch.float64, device=dev)
    prm = _params(cam_intr, cam_pose, None, dev)
    _lib.call("semabs_pointcloud_f64", _lib.ptr(d), H, W, _lib.ptr(prm), int(cam_pose is not None), _lib.ptr(xyz), _lib.stream())
    cam_pts = xyz.cpu().numpy()
    color_pts = None if color_img is None else color_img.reshape(-1, 3)
    return cam_pts, color_pts


def filter_pts_bounds(xyz, bounds):
    xyz = xyz.cpu().numpy() if isinstance(xyz, torch.Tensor) else np.asarray(xyz)
    b = np.asarray(bounds)
    m = np.ones(len(xyz), bool)
    for a in range(3):  # six compares: host glue, identical semantics to point_cloud.py:24-31
        m &= (xyz[:, a] >= b[0, a]) & (xyz[:, a] <= b[1, a])
    return m


def frustum_mask_device(pts64: torch.Tensor, h: int, w: int, cam_pose, cam_intr, prm: torch.Tensor | None = None) -> torch.Tensor:
    """pts64 f64 [M, 3] on the GPU -> uint8 [M] on the GPU (point_cloud.py:88-110); only the 16 pose / intrinsics doubles cross PCIe
    (prm: already there - `frustum_params`)."""
    dev = _lib.require_gpu()
    if prm is None:
        prm = frustum_params(cam_pose, cam_intr)
    pts64 = pts64.contiguous()
    mask = torch.empty(len(pts64), dtype=torch.uint8, device=dev)
    _lib.call("semabs_frustum_mask", _lib.ptr(pts64), len(pts64), _lib.ptr(prm), int(h), int(w), _lib.ptr(mask), _lib.stream())
    return mask


def check_pts_in_frustum(xyz_pts, depth, cam_pose, cam_intr):
    dev = _lib.require_gpu()
    pts = torch.from_numpy(np.ascontiguousarray(xyz_pts, dtype=np.float64)).to(dev)
    h, w = depth.shape
    return frustum_mask_device(pts, h, w, cam_pose, cam_intr).cpu().numpy().astype(bool)
