"""ORACLE (test infrastructure, not product code) — CPU restatement of the tile pre-processing.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Follows `CLIP/clip/clip_explainability.py:98-108` (`_transform`): Resize(224, BICUBIC) -> CenterCrop ->
RGB -> ToTensor -> Normalize, as applied to every square crop by `ClipWrapper.create_tiles`
(`CLIP/clip/__init__.py:276-280`).  The resize itself lives in a third-party dependency that is not under
/root/reference: Pillow (`semabs.yml:101` pins 9.2.0; this image has 12.2.0, same algorithm).  Its
published algorithm (src/libImaging/Resample.c: `precompute_coeffs`, `normalize_coeffs_8bpc`,
`ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`) is restated here in numpy:

  * per output index: center = (i + .5) * scale, support = 2 * max(scale, 1) (bicubic, a = -0.5),
    taps xmin..xmax, double-precision weights normalised to sum 1;
  * weights converted to 22-bit fixed point (round half away from zero);
  * horizontal pass then vertical pass, each: acc = 2^21 + sum(pixel * k) in int32, >> 22, clip to [0,255]
    (uint8 between the passes).

Pinned bit-exact against `PIL.Image.resize(..., BICUBIC)` of this image in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
OUT_RES = 224


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    r = np.zeros_like(x)
    m1 = x < 1.0
    m2 = (x >= 1.0) & (x < 2.0)
    r[m1] = ((a + 2.0) * x[m1] - (a + 3.0)) * x[m1] * x[m1] + 1
    r[m2] = (((x[m2] - 5) * x[m2] + 8) * x[m2] - 4) * a
    return r


def resample_coeffs(in_size: int, out_size: int):
    """-> (xmin[out], xcnt[out], kk[out, ksize] int32) exactly as Pillow's precompute + normalize."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xcnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = _bicubic((np.arange(n, dtype=np.float64) + lo - center + 0.5) * ss)
        ww = 0.0
        for v in w:  # sequential double sum, as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64),
                         (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        # C cast (int) truncates toward zero: astype does the same for these magnitudes
        xmin[xx], xcnt[xx] = lo, n
        kk[xx, :n] = fixed.astype(np.int32)
    return xmin, xcnt, kk


def _pass(img: np.ndarray, xmin, xcnt, kk, axis: int) -> np.ndarray:
    """One separable pass along `axis` of a uint8 [H, W, C] image."""
    src = np.moveaxis(img, axis, 0).astype(np.int32)  # [in, other, C]
    out_size, ksize = kk.shape
    idx = np.minimum(xmin[:, None] + np.arange(ksize)[None, :], src.shape[0] - 1)  # [out, ksize]
    gathered = src[idx]  # [out, ksize, other, C]   (taps beyond xcnt have k == 0)
    acc = (1 << (PRECISION_BITS - 1)) + np.einsum("ok,ok...->o...", kk.astype(np.int64), gathered.astype(np.int64))
    res = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(res, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_h: int = OUT_RES, out_w: int = OUT_RES) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [out_h, out_w, 3]; horizontal pass first (Pillow order)."""
    h, w = img.shape[:2]
    if w != out_w:
        img = _pass(img, *resample_coeffs(w, out_w), axis=1)
    if h != out_h:
        img = _pass(img, *resample_coeffs(h, out_h), axis=0)
    return img


def normalize_u8(img_u8: np.ndarray) -> np.ndarray:
    """uint8 [H, W, 3] -> fp32 [3, H, W]: ToTensor (/255) then Normalize, all fp32 like torchvision."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    mean = np.asarray(CLIP_MEAN, np.float32)[:, None, None]
    std = np.asarray(CLIP_STD, np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32)


def preprocess_tile(crop_u8: np.ndarray) -> np.ndarray:
    """Square uint8 crop [ts, ts, 3] -> fp32 [3, 224, 224] (CenterCrop(224) of a 224x224 image is the identity)."""
    assert crop_u8.shape[0] == crop_u8.shape[1], "tiles on the path are square"
    return normalize_u8(resize_bicubic_u8(crop_u8))
