"""Seeded synthetic scenes: there is no dataset here (matterport.png is a missing blob, the
269 GB THOR dataset cannot be fetched), so every test / bench input is generated from a seed.

RGB: uint8 smooth noise (4 octaves of bilinearly-upsampled uniform noise) so the bicubic resampler sees
non-trivial content; depth: fp32 in [1.5, 2.5] m, no zeros; intrinsics fx = fy = 400 * W/480, principal point at
the image centre; pose: the `arkit_vn_poster` extrinsics pattern (camera at (-2.056, -0.2, 1.901) looking along
+x, pitched down 0.3 rad) so most points fall inside scene_bounds [[-1,-1,-0.1],[1,1,1.9]].
"""
from __future__ import annotations

import numpy as np

SCENE_BOUNDS = ((-1.0, -1.0, -0.1), (1.0, 1.0, 1.9))


def _upsample_bilinear(a: np.ndarray, H: int, W: int) -> np.ndarray:
    h, w = a.shape[:2]
    ys = (np.arange(H) + 0.5) * h / H - 0.5
    xs = (np.arange(W) + 0.5) * w / W - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1); y1 = np.clip(y0 + 1, 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1); x1 = np.clip(x0 + 1, 0, w - 1)
    wy = np.clip(ys - y0, 0, 1)[:, None, None]
    wx = np.clip(xs - x0, 0, 1)[None, :, None]
    a = a.reshape(h, w, -1)
    top = a[y0][:, x0] * (1 - wx) + a[y0][:, x1] * wx
    bot = a[y1][:, x0] * (1 - wx) + a[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


def smooth_noise(H: int, W: int, C: int, rng: np.random.Generator) -> np.ndarray:
    """float64 [H, W, C] in [0, 1]."""
    acc = np.zeros((H, W, C))
    amp, tot = 1.0, 0.0
    for cells in (3, 7, 17, 41):
        acc += amp * _upsample_bilinear(rng.random((cells, cells, C)), H, W)
        tot += amp
        amp *= 0.5
    return acc / tot


def synth_rgb(H: int, W: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = smooth_noise(H, W, 3, rng)
    x = (x - x.min()) / (x.max() - x.min())
    return np.clip(np.round(x * 255), 0, 255).astype(np.uint8)


def synth_scene(H: int = 480, W: int = 480, seed: int = 0) -> dict:
    rng = np.random.default_rng(10_000 + seed)
    rgb = synth_rgb(H, W, seed)
    d = smooth_noise(H, W, 1, rng)[..., 0]
    d = (d - d.min()) / (d.max() - d.min())
    depth = (1.5 + 1.0 * d).astype(np.float32)
    f = 400.0 * W / 480.0
    K = np.array([[f, 0.0, W / 2.0], [0.0, f, H / 2.0], [0.0, 0.0, 1.0]])
    s, c = np.sin(0.3), np.cos(0.3)
    pose = np.array([[0.0, -s, c, -2.05570605], [-1.0, 0.0, 0.0, -0.2], [0.0, -c, -s, 1.90137071], [0.0, 0.0, 0.0, 1.0]])
    return dict(rgb=rgb, depth=depth, cam_intr=K, cam_pose=pose)


def synth_jitter(img: np.ndarray, k: int) -> np.ndarray:
    """Deterministic stand-in for one draw of the reference's ColorJitter (`CLIP/clip/__init__.py:55-57,246-247` is random):
    per-channel gain, brightness offset and a gamma curve chosen by `k`.  Parity fixtures inject these images on both sides
    (reference: `ClipWrapper.jittering_transforms`; here: `jittered_images=`), so the augmentation path - 6 images, 2 448
    forwards at 480 x 480 - is exercised with identical pixels."""
    rng = np.random.default_rng(777 + k)
    gain = rng.uniform(0.7, 1.3, size=3)
    off = rng.uniform(-20, 20)
    gamma = rng.uniform(0.7, 1.4)
    x = img.astype(np.float64) / 255.0
    x = np.clip(x, 0, 1) ** gamma * gain[None, None, :] + off / 255.0
    return np.clip(np.round(x * 255.0), 0, 255).astype(np.uint8)


def synth_ovssc_logits(points, n_classes: int):
    """Closed-form stand-in for the network inside `process_batch_ovssc` (parity fixture g18): fp32 multiply / add only (IEEE-exact, identical
    on any host or device), values straddling the -3 cutoff with changing arg-max regions.  points torch fp32 [M, 3] -> [n_classes, M]."""
    import torch
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    out = []
    for c in range(n_classes):
        a, b, d = 1.0 + 0.5 * c, 0.75 - 0.25 * c, 0.5 + 0.125 * c
        q = x * a + y * b
        q = q + z * d
        q = q + (0.25 * c - 0.5)
        out.append(q * q * (-2.0) + (0.5 - 0.375 * c) + x * (c - 1.5))
    return torch.stack(out, dim=0)


VOOL_RELATIONS = ["in", "behind", "in front of", "on the left of", "on the right of", "on", "[pad]"]


def synth_vool_logits(points, target_first: float, reference_first: float, relation: str):
    """Closed-form stand-in for the network inside `process_batch_vool` (parity fixture g26): a pure fp32 multiply / add function of the query points
    and of WHAT the caller selected for this description - the first target / reference saliency value it passed and the relation name - so that the
    reference's plumbing (per-description row selection, relation list nesting, 2^k chunking with a ragged tail, concatenation, return form) can be
    compared bit for bit.  points torch fp32 [M, 3] -> [M]."""
    r = float(VOOL_RELATIONS.index(relation))
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    q = x * (1.0 + 0.25 * r) + y * (0.5 - 0.125 * r)
    q = q + z * 0.75
    return q * float(target_first) + (q * q) * float(reference_first) + (r - 2.0)


def synth_relevancy(img: np.ndarray, labels) -> np.ndarray:
    """Closed-form stand-in for `ClipWrapper.get_clip_saliency(...)[0]` (parity fixture g25: `prep_data` executed from the reference's source):
    one fp32 map [H, W] per label STRING, a pure function of the label's bytes and the image - so that the reference's data plumbing around
    the CLIP call (key set, x 50, mean subtraction, in-bounds selection, stacking) can be compared bit for bit.  -> fp32 [L, H, W]."""
    import zlib
    H, W = img.shape[:2]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    out = []
    for lab in labels:
        h = zlib.crc32(str(lab).encode())
        a, b, c = np.float32((h & 0xff) / 256.0), np.float32(((h >> 8) & 0xff) / 256.0), int((h >> 16) & 3) % 3
        m = img[..., c].astype(np.float32) * np.float32(1.0 / 255.0) * a + (yy * np.float32(1.0 / 64.0) - xx * np.float32(1.0 / 128.0)) * b
        out.append((m * np.float32(0.004)).astype(np.float32))
    return np.stack(out, axis=0)
