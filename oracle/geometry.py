"""ORACLE (test infrastructure, not product code) — CPU restatement of the geometry half of the path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

  get_pointcloud / transform_pointcloud   point_cloud.py:34-66, 8-21
  filter_pts_bounds                       point_cloud.py:24-31
  check_pts_in_frustum                    point_cloud.py:88-110
  VirtualGrid.get_points_grid_idxs / flatten_idxs   net.py:84-133
  TSDFVolume.__init__/vox2world/cam2pix/integrate_tsdf/integrate   fusion.py:37-195

dtype contract (as the reference behaves when imported in this image — numpy 2.2):
`depth / fx` is f32-array / f64-scalar = f64 (NEP 50); voxel centres are f32(f64(origin) + f64(vs) * idx)
(numba types the python-float voxel size as float64); the pose transform is f64; intrinsics are cast to
f32 before `x * fx / z + cx` (f64 result); pixel indices are round-half-to-even of that.
"""
from __future__ import annotations

import numpy as np


# ---- point cloud ---------------------------------------------------------------------------------
def get_pointcloud(depth: np.ndarray, cam_intr: np.ndarray, cam_pose: np.ndarray | None) -> np.ndarray:
    """fp32 [H, W] depth -> f64 [H*W, 3] world points (point_cloud.py:51-66)."""
    h, w = depth.shape
    px, py = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    fx, fy = np.float64(cam_intr[0, 0]), np.float64(cam_intr[1, 1])
    cx, cy = np.float64(cam_intr[0, 2]), np.float64(cam_intr[1, 2])
    d = depth.astype(np.float64)
    x = (px - cx) * (d / fx)
    y = (py - cy) * (d / fy)
    pts = np.stack([x, y, d], axis=-1).reshape(-1, 3)
    if cam_pose is not None:
        R, t = np.asarray(cam_pose, np.float64)[:3, :3], np.asarray(cam_pose, np.float64)[:3, 3]
        pts = (R @ pts.T).T + t[None, :]
    return pts


def filter_pts_bounds(xyz: np.ndarray, bounds: np.ndarray) -> np.ndarray:
    m = np.ones(len(xyz), bool)
    for a in range(3):
        m &= (xyz[:, a] >= bounds[0, a]) & (xyz[:, a] <= bounds[1, a])
    return m


def check_pts_in_frustum(xyz: np.ndarray, depth_shape, cam_pose: np.ndarray, cam_intr: np.ndarray) -> np.ndarray:
    T = np.linalg.inv(cam_pose)
    cam = (T[:3, :3] @ np.asarray(xyz, np.float64).T).T + T[:3, 3][None]
    z = cam[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        px = (cam_intr[0, 0] / z) * cam[:, 0] + cam_intr[0, 2]
        py = (cam_intr[1, 1] / z) * cam[:, 1] + cam_intr[1, 2]
    h, w = depth_shape
    return (px >= 0) & (px < w) & (py >= 0) & (py < h) & (z > 0)


# ---- voxel indices -------------------------------------------------------------------------------
def grid_constants(scene_bounds, grid_shape):
    """(offsets = -lc, scales = (S-1)/(uc-lc)) as fp32, formed exactly as net.py:91-95."""
    lc = np.asarray(scene_bounds[0], np.float64).astype(np.float32)
    uc = np.asarray(scene_bounds[1], np.float64).astype(np.float32)
    idx_scale = np.asarray(grid_shape, np.float32) - np.float32(1)
    return (-lc).astype(np.float32), (idx_scale / (uc - lc)).astype(np.float32)


def points_grid_idxs(points: np.ndarray, scene_bounds, grid_shape) -> np.ndarray:
    """fp32 [..., 3] -> int64 [..., 3]: trunc((p + (-lc)) * scales) clamped to [0, S-1]."""
    off, sc = grid_constants(scene_bounds, grid_shape)
    f = (points.astype(np.float32) + off) * sc           # two separately-rounded fp32 ops
    i = np.trunc(f).astype(np.int64)
    return np.clip(i, 0, np.asarray(grid_shape, np.int64) - 1)


def flatten_idxs(idxs: np.ndarray, grid_shape) -> np.ndarray:
    S0, S1, S2 = grid_shape
    return idxs[..., 0] * (S1 * S2) + idxs[..., 1] * S2 + idxs[..., 2]


# ---- TSDF ----------------------------------------------------------------------------------------
class TSDFVolume:
    """Vectorised numpy restatement of fusion.py:37-195 (TSDF + weight + packed colour volumes)."""

    def __init__(self, vol_bnds, voxel_size):
        vol_bnds = np.array(vol_bnds, dtype=np.float64)
        self._voxel_size = float(voxel_size)
        self._trunc_margin = 5 * self._voxel_size
        self._vol_dim = np.ceil((vol_bnds[:, 1] - vol_bnds[:, 0]) / self._voxel_size).astype(int)
        self._vol_origin = vol_bnds[:, 0].astype(np.float32)
        self._tsdf_vol_cpu = -np.ones(self._vol_dim, np.float32)
        self._weight_vol_cpu = np.zeros(self._vol_dim, np.float32)
        self._color_vol_cpu = np.zeros(self._vol_dim, np.float32)
        g = np.meshgrid(*[np.arange(d) for d in self._vol_dim], indexing="ij")
        self.vox_coords = np.stack([a.reshape(-1) for a in g], axis=1).astype(int)

    def project(self, cam_intr, cam_pose):
        """-> (pix int64 [N, 2], pix_z f64 [N]) : vox2world + rigid_transform + cam2pix."""
        world = (self._vol_origin.astype(np.float64)[None]
                 + np.float64(self._voxel_size) * self.vox_coords.astype(np.float32).astype(np.float64))
        world = world.astype(np.float32)
        T = np.linalg.inv(np.asarray(cam_pose, np.float64))
        xyz_h = np.hstack([world, np.ones((len(world), 1), np.float32)])
        cam = np.dot(T, xyz_h.T).T[:, :3]
        intr = np.asarray(cam_intr).astype(np.float32)
        fx, fy, cx, cy = intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            px = np.round(cam[:, 0] * np.float64(fx) / cam[:, 2] + np.float64(cx))
            py = np.round(cam[:, 1] * np.float64(fy) / cam[:, 2] + np.float64(cy))
        return np.stack([px, py], axis=1).astype(np.int64), cam[:, 2]

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.0):
        im_h, im_w = depth_im.shape
        color = color_im.astype(np.float32)
        color = np.floor(color[..., 2] * np.float32(256 * 256) + color[..., 1] * np.float32(256) + color[..., 0])
        pix, pix_z = self.project(cam_intr, cam_pose)
        px, py = pix[:, 0], pix[:, 1]
        valid_pix = (px >= 0) & (px < im_w) & (py >= 0) & (py < im_h) & (pix_z > 0)
        depth_val = np.zeros(len(px))
        depth_val[valid_pix] = depth_im[py[valid_pix], px[valid_pix]]
        diff = depth_val - pix_z
        valid = (depth_val > 0) & (diff >= -self._trunc_margin)
        dist = np.maximum(-1, np.minimum(1, diff / self._trunc_margin))
        vx, vy, vz = (self.vox_coords[valid, i] for i in range(3))
        w_old = self._weight_vol_cpu[vx, vy, vz]
        tsdf = self._tsdf_vol_cpu[vx, vy, vz]
        ow = np.float64(obs_weight)
        w_new = (w_old.astype(np.float64) + ow).astype(np.float32)
        tsdf_new = (((w_old * tsdf).astype(np.float64) + ow * dist[valid]) / w_new.astype(np.float64)).astype(np.float32)
        self._weight_vol_cpu[vx, vy, vz] = w_new
        self._tsdf_vol_cpu[vx, vy, vz] = tsdf_new
        # colour (fp32 throughout, fusion.py:176-195)
        cc = np.float32(256 * 256)
        old = self._color_vol_cpu[vx, vy, vz]
        ob = np.floor(old / cc); og = np.floor((old - ob * cc) / np.float32(256)); orr = old - ob * cc - og * np.float32(256)
        new = color[py[valid], px[valid]]
        nb = np.floor(new / cc); ng = np.floor((new - nb * cc) / np.float32(256)); nr = new - nb * cc - ng * np.float32(256)
        owf = np.float32(obs_weight)
        nb = np.minimum(np.float32(255), np.round((w_old * ob + owf * nb) / w_new))
        ng = np.minimum(np.float32(255), np.round((w_old * og + owf * ng) / w_new))
        nr = np.minimum(np.float32(255), np.round((w_old * orr + owf * nr) / w_new))
        self._color_vol_cpu[vx, vy, vz] = nb * cc + ng * np.float32(256) + nr
        self._last = dict(pix=pix, valid=valid)
