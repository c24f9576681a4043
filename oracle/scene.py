"""ORACLE (test infrastructure, not product code) — CPU restatement of the end-to-end OVSSC recipe
(visualize.prep_data + process_batch_ovssc, visualize.py:61-154, 157-248) with the feature volume computed once per
label (same function of the same sub-sample as the reference's chunked loop).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch

from . import geometry as og
from . import relevancy as orl
from . import semabs3d as os3


def grid_points(scene_bounds, S):
    lc = np.asarray(scene_bounds[0], np.float32)
    uc = np.asarray(scene_bounds[1], np.float32)
    scales = (uc - lc) / (np.asarray([S, S, S], np.float32) - np.float32(1))
    g = np.stack(np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij"), axis=-1).astype(np.float32)
    return (g * scales + lc).reshape(-1, 3).astype(np.float32)


def sample_points(sampling_shape, scene_bounds):
    """visualize.get_sample_points (:283-298): idx * ((uc - lc) / (shape - 1)) + lc in fp32, C order."""
    lc = np.asarray(scene_bounds[0], np.float32)
    uc = np.asarray(scene_bounds[1], np.float32)
    scales = (uc - lc) / (np.asarray(sampling_shape, np.float32) - np.float32(1))
    g = np.stack(np.meshgrid(*[np.arange(s) for s in sampling_shape], indexing="ij"), axis=-1).astype(np.float32)
    return (g * scales + lc).reshape(-1, 3).astype(np.float32)


def ovssc_post_mask(logits: torch.Tensor, scene, scene_bounds, sampling_shape, cutoff=-3.0):
    """The tail of visualize.process_batch_ovssc (:212-248): TSDF at the sampling resolution, arg-max over classes, "all below cutoff" /
    out-of-frustum / tsdf > 0 masks -> fp32 {0, 1} volumes [C, *sampling_shape].  logits [C, prod(shape)].  Pinned by g18."""
    H, W = scene["depth"].shape
    lo, hi = np.asarray(scene_bounds[0], np.float64), np.asarray(scene_bounds[1], np.float64)
    tv = og.TSDFVolume(np.stack([lo, hi], axis=1), (scene_bounds[1][0] - scene_bounds[0][0]) / sampling_shape[0])
    tv.integrate(scene["rgb"], scene["depth"], scene["cam_intr"], scene["cam_pose"])
    q = sample_points(sampling_shape, scene_bounds)
    fr = og.check_pts_in_frustum(q.astype(np.float64), (H, W), scene["cam_pose"], scene["cam_intr"])
    arg = logits.argmax(dim=0).numpy()
    empty = (logits < cutoff).all(dim=0).numpy() | ~fr | (tv._tsdf_vol_cpu.reshape(-1) > 0)
    C = logits.shape[0]
    vols = np.stack([((arg == c) & ~empty) for c in range(C)]).astype(np.float32)
    return vols.reshape(C, *sampling_shape)


def subsample_indices(seed: int, n_in: int, num: int) -> np.ndarray:
    """The seeded draw with replacement standing in for visualize.py:193 `np.random.choice(len(pts), size=num_input_pts)` (unseeded in the
    reference): index j = mulhi64(splitmix64(seed * 0x9E3779B97F4A7C15 + j), n_in), a counter-based generator the device evaluates without
    knowing n_in on the host (csrc/geometry.hip k_subsample)."""
    M = (1 << 64) - 1
    out = np.empty(num, np.int64)
    base = (int(seed) * 0x9E3779B97F4A7C15) & M
    for j in range(num):
        x = (base + j + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        x ^= x >> 31
        out[j] = (x * int(n_in)) >> 64
    return out


def run_scene(clip_sd, net_sd, scene, w_text, scene_bounds, S, num_input_pts, seed, cfg, images=None, subtract_mean=True,
              cutoff=-3.0, with_tsdf=True):
    """-> dict(relevancies [L,H,W] (x50, mean-subtracted), logits [L, S^3], labels [S^3], tsdf [S,S,S])."""
    H, W = scene["depth"].shape
    imgs = images if images is not None else [scene["rgb"]]
    with torch.no_grad():
        maps = orl.relevancy_maps(clip_sd, imgs, w_text.T.contiguous(), **cfg)        # w_text given as [L, E]
    rel = maps * 50
    pts = og.get_pointcloud(scene["depth"], scene["cam_intr"], scene["cam_pose"]).astype(np.float32)
    mask = og.filter_pts_bounds(pts, np.asarray(scene_bounds, np.float64))
    if subtract_mean:
        rel = rel - rel.mean(dim=0, keepdim=True)
    pix = np.nonzero(mask)[0]
    sel = pix[subsample_indices(seed, len(pix), num_input_pts)]
    feat = rel.reshape(rel.shape[0], -1)[:, sel]                                           # [L, n]
    xyz = torch.from_numpy(pts[sel])[None]
    q = torch.from_numpy(grid_points(scene_bounds, S))
    L = feat.shape[0]
    with torch.no_grad():
        logits = os3.semabs3d_forward(net_sd, xyz, feat[None, :, :, None], q[None, None].repeat(1, L, 1, 1), scene_bounds, (S, S, S))[0]
    out = dict(relevancies=rel, logits=logits, n_in_bounds=len(pix))
    if with_tsdf:
        lo, hi = np.asarray(scene_bounds[0], np.float64), np.asarray(scene_bounds[1], np.float64)
        tv = og.TSDFVolume(np.stack([lo, hi], axis=1), (hi[0] - lo[0]) / S)
        tv.integrate(scene["rgb"], scene["depth"], scene["cam_intr"], scene["cam_pose"])
        fr = og.check_pts_in_frustum(q.numpy().astype(np.float64), (H, W), scene["cam_pose"], scene["cam_intr"])
        best, arg = logits.max(dim=0)
        empty = (logits < cutoff).all(dim=0).numpy() | ~fr | (tv._tsdf_vol_cpu.reshape(-1) > 0)
        labels = arg.numpy().astype(np.int32)
        labels[empty] = -1
        out.update(labels=labels, tsdf=tv._tsdf_vol_cpu)
    return out
