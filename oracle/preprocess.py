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


# ----------------------------------------------------------------------------------------------------------------------
# Colour jitter of the augmentation copies: `ClipWrapper.jittering_transforms = ColorJitter(brightness=0.6, contrast=0.6,
# saturation=0.6, hue=0.1)` applied to the PIL image (`CLIP/clip/__init__.py:55-57, 246-247`).  The arithmetic lives in two
# third-party dependencies absent from /root/reference: torchvision 0.13.1 (`semabs.yml:129`: transforms.ColorJitter.forward ->
# functional_pil.adjust_brightness / adjust_contrast / adjust_saturation / adjust_hue, applied in the order of a random permutation
# with factors U(0.4, 1.6)^3, U(-0.1, 0.1)) and Pillow 9.2.0 (`semabs.yml:101`; this image: 12.2.0): ImageEnhance.Brightness /
# Contrast / Color = Image.blend(degenerate, image, factor) (libImaging/Blend.c), convert("L") (Convert.c rgb2l: ITU-R 601-2 luma
# in 16.16 fixed point), convert("HSV") / convert("RGB") (Convert.c rgb2hsv_row / hsv2rgb: uint8-quantised H, S, V).  Restated here
# in numpy; pinned BIT-EXACT against this image's Pillow: both HSV conversions on all 2^24 colours, the blends on all 2^24 colours
# at several factors (tests/golden/gen_golden.py g28 -> tests/test_oracle_golden.py).  The random draw itself is not part of the
# contract (the reference's is torch's global generator): (order, factors) are arguments.
#   op ids as in torchvision's fn_idx: 0 brightness, 1 contrast, 2 saturation, 3 hue.
# ----------------------------------------------------------------------------------------------------------------------
def _grey_u8(img: np.ndarray) -> np.ndarray:
    """convert("L"): (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16."""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def _blend_u8(degenerate: np.ndarray, img: np.ndarray, factor: float) -> np.ndarray:
    """Image.blend(degenerate, img, factor): fp32 `d + alpha * (x - d)` (alpha is a C float), truncated; clipped to [0, 255] first when
    alpha is outside [0, 1] (for alpha inside, the value already is)."""
    a = np.float32(factor)
    d, x = degenerate.astype(np.float32), img.astype(np.float32)
    t = (d + (a * (x - d)).astype(np.float32)).astype(np.float32)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t)).astype(np.int32).astype(np.uint8)


def jitter_brightness(img: np.ndarray, f: float) -> np.ndarray:
    return _blend_u8(np.zeros_like(img), img, f)


def jitter_contrast(img: np.ndarray, f: float) -> np.ndarray:
    g = _grey_u8(img)
    mean = int(float(g.astype(np.int64).sum()) / g.size + 0.5)           # int(ImageStat.Stat(L).mean[0] + 0.5)
    return _blend_u8(np.full_like(img, mean), img, f)


def jitter_saturation(img: np.ndarray, f: float) -> np.ndarray:
    return _blend_u8(np.repeat(_grey_u8(img)[..., None], 3, axis=2), img, f)


def rgb_to_hsv_u8(img: np.ndarray) -> np.ndarray:
    """Pillow convert("HSV"): fp32 s, rc, gc, bc; h = 2 + rc - bc etc. in double rounded to fp32; fmod(h / 6 + 1, 1) in double; truncation to uint8."""
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    maxc = np.maximum(r, np.maximum(g, b)).astype(np.int32)
    minc = np.minimum(r, np.minimum(g, b)).astype(np.int32)
    ok = maxc != minc
    cr = (maxc - minc).astype(np.float32)
    crs = np.where(ok, cr, np.float32(1))
    s = (cr / np.where(maxc > 0, maxc, 1).astype(np.float32)).astype(np.float32)
    rc = ((maxc - r).astype(np.float32) / crs).astype(np.float32)
    gc = ((maxc - g).astype(np.float32) / crs).astype(np.float32)
    bc = ((maxc - b).astype(np.float32) / crs).astype(np.float32)
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    return np.stack([np.where(ok, uh, 0), np.where(ok, us, 0), maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb_u8(hsv: np.ndarray) -> np.ndarray:
    """Pillow HSV -> RGB: i = floor(h * 6 / 255), f (fp32) its remainder, p / q / t = round-half-away(v * (1 - ...)) in double."""
    c_round = lambda x: np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))
    s, v = hsv[..., 1], hsv[..., 2]
    hd = hsv[..., 0].astype(np.float64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int32)
    f = (hd - i.astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vf = v.astype(np.float64)
    p = np.clip(c_round(vf * (1.0 - fs)), 0, 255).astype(np.int32)
    q = np.clip(c_round(vf * (1.0 - fs * f)), 0, 255).astype(np.int32)
    t = np.clip(c_round(vf * (1.0 - fs * (1.0 - f))), 0, 255).astype(np.int32)
    vi = v.astype(np.int32)
    sec = i % 6
    out = np.stack([np.choose(sec, [vi, q, p, p, t, vi]), np.choose(sec, [t, vi, vi, q, p, p]), np.choose(sec, [p, p, t, vi, vi, q])], axis=-1)
    return np.where((s == 0)[..., None], vi[..., None], out).astype(np.uint8)


def hue_shift_u8(f: float) -> int:
    """torchvision functional_pil.adjust_hue: `np_h += np.uint8(hue_factor * 255)` - truncation toward zero, then uint8 wrap-around
    (numpy 1.22, the reference's pin, wraps a negative value; numpy 2 raises - the wrap is the contract)."""
    return int(f * 255) % 256


def jitter_hue(img: np.ndarray, f: float) -> np.ndarray:
    hsv = rgb_to_hsv_u8(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(f)).astype(np.uint8)
    return hsv_to_rgb_u8(hsv)


JITTER_OPS = (jitter_brightness, jitter_contrast, jitter_saturation, jitter_hue)


def color_jitter(img: np.ndarray, order, factors) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [H, W, 3]: the four adjustments in `order` (a permutation of op ids), `factors[op]` each (ColorJitter.forward)."""
    out = np.ascontiguousarray(img)
    for op in order:
        out = JITTER_OPS[int(op)](out, float(factors[int(op)]))
    return out
