"""Import the read-only reference (`/root/reference`) in THIS container to generate golden vectors.

Used only by `tests/golden/gen_golden.py` (run once, here; never on the GPU box, which has no
`/root/reference`).  The reference is pure Python but needs packages this image lacks
(torchvision, ftfy, torch_scatter, torchtyping, typeguard, numba, skimage, pybullet, ...).  The
stand-ins below are OUR code and define the contract the fixtures pin:

* `torchvision.transforms`: PIL-based `Resize` (`Image.resize(..., BICUBIC)` = what torchvision does
  for PIL inputs), `CenterCrop`, `ToTensor` (`uint8 -> float32 / 255`), `Normalize`
  (`(x - mean) / std` in fp32), identity `ColorJitter` (the real one is random; parity runs use
  `augmentations=0` or this identity jitter and say so).
* `torch_scatter.scatter`: zeros-init + index_add / count for "mean" (empty voxels = 0), amax for "max".
* `numba.njit/prange`: identity / range  (python semantics of the same loops; scalar python floats are
  made explicit float64 by the callers as numba would type them).
* `ftfy.fix_text`: identity (labels are ASCII); `torchtyping`, `typeguard`: no-ops.

No reference source is copied; this file only arranges for the unmodified files to import.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if "torchvision" in sys.modules and getattr(sys.modules["torchvision"], "_semabs_stub", False):
        return

    # ---- torchvision.transforms (PIL path) -------------------------------------------------
    class InterpolationMode:
        BICUBIC = Image.BICUBIC
        BILINEAR = Image.BILINEAR
        NEAREST = Image.NEAREST

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=Image.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            w, h = img.size
            if isinstance(self.size, int):
                short, long = (w, h) if w <= h else (h, w)
                if short == self.size:
                    return img
                new_short, new_long = self.size, int(self.size * long / short)
                nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            else:
                nh, nw = self.size
            return img.resize((nw, nh), self.interpolation)

    class CenterCrop:
        def __init__(self, size):
            self.size = (size, size) if isinstance(size, int) else size

        def __call__(self, img):
            w, h = img.size
            th, tw = self.size
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
            return img.crop((left, top, left + tw, top + th))

    class ToTensor:
        def __call__(self, pic):
            arr = np.array(pic, dtype=np.uint8, copy=True)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            t = torch.from_numpy(arr).permute(2, 0, 1).contiguous()
            return t.to(dtype=torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            t = t.clone()
            mean = torch.as_tensor(self.mean, dtype=t.dtype)
            std = torch.as_tensor(self.std, dtype=t.dtype)
            return t.sub_(mean[:, None, None]).div_(std[:, None, None])

    class ColorJitter:
        def __init__(self, *a, **k):
            pass

        def __call__(self, img):
            return img

    tv = _mod("torchvision", _semabs_stub=True)
    tv.transforms = _mod(
        "torchvision.transforms", Compose=Compose, Resize=Resize, CenterCrop=CenterCrop,
        ToTensor=ToTensor, Normalize=Normalize, ColorJitter=ColorJitter,
        InterpolationMode=InterpolationMode,
    )

    # ---- ftfy, torchtyping, typeguard --------------------------------------------------------
    _mod("ftfy", fix_text=lambda s: s)

    class _TT:
        def __class_getitem__(cls, item):
            return torch.Tensor

    _mod("torchtyping", TensorType=_TT, patch_typeguard=lambda: None)
    _mod("typeguard", typechecked=lambda f=None, **k: f if f is not None else (lambda g: g))

    # ---- torch_scatter ------------------------------------------------------------------------
    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        assert dim in (-2, src.dim() - 2)
        B, N, C = src.shape
        dim_size = int(dim_size)
        idx = index.unsqueeze(-1).expand(B, N, C)
        if reduce in ("sum", "add", "mean"):
            res = torch.zeros(B, dim_size, C, dtype=src.dtype, device=src.device)
            res.scatter_add_(1, idx, src)
            if reduce == "mean":
                cnt = torch.zeros(B, dim_size, dtype=src.dtype, device=src.device)
                cnt.scatter_add_(1, index, torch.ones_like(index, dtype=src.dtype))
                res = res / cnt.clamp(min=1).unsqueeze(-1)
            return res
        if reduce == "max":
            res = torch.zeros(B, dim_size, C, dtype=src.dtype, device=src.device)
            return res.scatter_reduce(1, idx, src, reduce="amax", include_self=False)
        raise NotImplementedError(reduce)

    _mod("torch_scatter", scatter=scatter)

    # ---- numba, skimage, pybullet --------------------------------------------------------------
    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _mod("numba", njit=njit, prange=range)
    sk = _mod("skimage")
    sk.measure = _mod("skimage.measure")
    _mod("pybullet")
    _mod("pybullet_data")
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mpl = _mod("matplotlib")
        mpl.pyplot = _mod("matplotlib.pyplot")


_CKPT = {}


def checkpoint_path(arch: str, seed: int, sharpen: float = 2.0, stats: str = "init") -> str:
    """Save our seeded state dict once to a temp .pt the reference `load()` can read."""
    import semabs_amd  # noqa: F401
    from semabs_amd.weights import make_clip_state_dict

    key = (arch, seed, sharpen, stats)
    if key not in _CKPT:
        sd = make_clip_state_dict(arch, seed, sharpen, stats=stats)
        # the reference derives the architecture from these (model_explainability.py:530-594)
        path = os.path.join(tempfile.gettempdir(), f"semabs_clip_{arch.replace('/', '-')}_{seed}_{sharpen}_{stats}.pt")
        torch.save(sd, path)
        _CKPT[key] = path
    return _CKPT[key]


class TileList:
    """numpy-2 shim for `tiles = np.array(tiles)` of slice tuples (`CLIP/clip/__init__.py:282`):
    supports the bool-mask and slice indexing `get_clip_saliency_convolve` does (`:210,223`)."""

    def __init__(self, items):
        self.items = list(items)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return TileList(self.items[k])
        if isinstance(k, np.ndarray) and k.dtype == bool:
            return TileList([t for t, m in zip(self.items, k) if m])
        return self.items[k]

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


def load_reference_clip(arch: str = "ViT-B/32", seed: int = 0, sharpen: float = 2.0, stats: str = "init"):
    """Return the reference `CLIP.clip` package with ClipWrapper initialised on our seeded weights."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import CLIP.clip as rc
    import CLIP.clip.clip as rclip
    import CLIP.clip.clip_explainability as rexp

    path = checkpoint_path(arch, seed, sharpen, stats)
    rclip._download = lambda url, root=None: path
    rexp._download = lambda url, root=None: path
    W = rc.ClipWrapper
    W.clip_model = W.clip_preprocess = W.clip_gradcam = None
    W.check_initialized(clip_model_type=arch)

    orig = W.create_tiles.__func__

    def create_tiles(cls, *a, **k):
        tiles, tile_imgs, counts, tile_sizes = orig(cls, *a, **k)
        return TileList([tuple(t) for t in tiles]), tile_imgs, counts, tile_sizes

    # np.array(list of slice tuples) is an object array; indexing a tensor with its rows fails on
    # torch 2.x, so hand back a list-like with the same element order (SURVEY.md §8c shim i).
    import numpy as _np
    _orig_array = _np.array

    def create_tiles_wrapped(cls, img, augmentations, cropping_augmentations, **kwargs):
        return create_tiles(cls, img=img, augmentations=augmentations,
                            cropping_augmentations=cropping_augmentations, **kwargs)

    W.create_tiles = classmethod(create_tiles_wrapped)
    return rc


def load_reference_geometry():
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import fusion
    import point_cloud
    return fusion, point_cloud


def load_reference_net():
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # net.py imports CLIP.clip (for baselines we do not use); that import needs the stubs only
    import net
    import unet3d
    return net, unet3d
