"""Drop-in for the relevancy surface of the reference's `CLIP.clip` package: `ClipWrapper`, `saliency_configs`, `imagenet_templates`
(`from CLIP.clip import ClipWrapper, saliency_configs, imagenet_templates`, generate_relevancy.py:10).  `imagenet_prompt_ensemble` is a flag
for the CALLER (generate_relevancy.py:70-80 passes `imagenet_templates` as `prompts` when a config sets it; both shipped configs leave it
False); `get_clip_saliency` itself takes any template list: the zero-shot weight of a label is the mean over the templates' normalised
text features (clip_gradcam.py:12-27).

    ClipWrapper.get_clip_saliency(img, text_labels, prompts, **saliency_configs["ours"](h))
        -> (fp32 [L, H, W] on CPU, text features [L, 512] on CPU)            CLIP/clip/__init__.py:103-133

Everything between the uint8 image and the fp32 maps runs in libsemabs_hip.so:
tile crop + Pillow-exact bicubic 224 + normalise + im2col (tiles.hip) -> ViT trunk / last-block CLS path / analytic
rollout (gemm.hip, vit.hip) -> un-flip average + bilinear upsample + fp16 canvases + count normalise (tiles.hip).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .templates import imagenet_templates
from .vit import TextEncoder, VisionRollout, make_vision_engine

KMAX = 24
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

saliency_configs = {
    "ours": lambda img_dim: {
        "distractor_labels": {},
        "horizontal_flipping": True,
        "augmentations": 5,
        "imagenet_prompt_ensemble": False,
        "positive_attn_only": True,
        "cropping_augmentations": [
            {"tile_size": img_dim, "stride": img_dim // 4},
            {"tile_size": int(img_dim * 2 / 3), "stride": int(img_dim * 2 / 3) // 4},
            {"tile_size": img_dim // 2, "stride": (img_dim // 2) // 4},
            {"tile_size": img_dim // 4, "stride": (img_dim // 4) // 4},
        ],
    },
    "chefer_et_al": lambda img_dim: {
        "distractor_labels": {},
        "horizontal_flipping": False,
        "augmentations": 0,
        "imagenet_prompt_ensemble": False,
        "positive_attn_only": True,
        "cropping_augmentations": [{"tile_size": img_dim, "stride": img_dim // 4}],
    },
}


def plan_tiles(H: int, W: int, n_images: int, cropping_augmentations):
    """Tile geometry in the reference's append order (CLIP/clip/__init__.py:257-274; rows are called x, columns y,
    and the loop bounds compare the column start against H and the row start against W — kept as is).
    -> (table int32 [N, 4] = (image, row0, col0, tile_size), scales int32 [S, 5] = (ts, stride, n_rows, n_cols, base))"""
    scales, per_img, base = [], [], 0
    seen = set()
    for aug in cropping_augmentations:
        ts, stride = int(aug["tile_size"]), int(aug["stride"])
        if ts in seen:
            raise NotImplementedError("two cropping_augmentations with the same tile_size share one canvas in the "
                                      "reference; not supported")
        seen.add(ts)
        ys = [int(y) for y in np.arange(0, W - ts + 1, stride) if y < H]
        xs = [int(x) for x in np.arange(0, H - ts + 1, stride) if x < W]
        scales.append((ts, stride, len(xs), len(ys), base))
        for y in ys:
            for x in xs:
                per_img.append((x, y, ts))
        base += len(xs) * len(ys)
    rows = [(im, x, y, ts) for im in range(n_images) for (x, y, ts) in per_img]
    return (np.asarray(rows, np.int32).reshape(-1, 4), np.asarray(scales, np.int32).reshape(-1, 5))


class _ResizeCoeffs:
    """Per tile size: Pillow's fixed-point bicubic taps (host arithmetic in libsemabs_hip.so), cached on the GPU."""

    def __init__(self, dev):
        self.dev = dev
        self.sizes: List[int] = []
        self.xmin: List[np.ndarray] = []
        self.kk: List[np.ndarray] = []
        self.ksize: List[int] = []
        self._dev = None

    def id_of(self, ts: int) -> int:
        if ts not in self.sizes:
            xmin = np.zeros(224, np.int32)
            kk = np.zeros((224, KMAX), np.int32)
            ks = C.c_int(0)
            _lib.call("semabs_resize_coeffs", int(ts), 224, xmin.ctypes.data, kk.ctypes.data, KMAX, C.addressof(ks))
            self.sizes.append(ts); self.xmin.append(xmin); self.kk.append(kk); self.ksize.append(int(ks.value))
            self._dev = None
        return self.sizes.index(ts)

    def device(self):
        if self._dev is None:
            self._dev = (torch.from_numpy(np.stack(self.xmin)).to(self.dev), torch.from_numpy(np.stack(self.kk)).to(self.dev),
                         torch.tensor(self.ksize, dtype=torch.int32, device=self.dev))
        return self._dev


class ClipWrapper:
    """Class-level singleton with the reference's classmethod surface."""

    engine: Optional[VisionRollout] = None
    text: Optional[TextEncoder] = None
    tokenizer = None
    device = None
    clip_model_type = None
    templates: List[str] = ["{}"]
    positive_attn_only = False
    class_to_language_feature: Dict[str, torch.Tensor] = {}
    _coeffs: Optional[_ResizeCoeffs] = None
    _lut = None
    _rng = np.random.default_rng(0)
    n_streams = 1                       # HIP streams the independent tile chunks are pipelined over (when a scene is cut into several batches)
    _streams = None
    _patches = {}
    _plans = {}                         # (H, W, n_img, cropping_augmentations) -> tile table / scale table / per-tile coefficient ids, host and device copies
    _jitter_scratch = None              # 8 bytes of device memory for semabs_color_jitter (the grey-level sum of the contrast step)
    state_dict_provider = None          # callable(clip_model_type) -> state dict; set by tests / bench

    # ---- initialisation ----------------------------------------------------------------------------
    def __init__(self, clip_model_type, device=None, state_dict=None, chunk_tiles=2448, max_labels=32, precision: str | None = None, **kwargs):
        """precision: None / "default" = fp16 MFMA operands throughout the trunk (the reference's own CUDA dtype); "parity" = q and k additionally carried
        as fp16 hi + lo pairs into the attention scores (ViT-B; ~4 % slower, relevancy maps ~30 % closer to the fp32 CPU reference - clip/vit.py)."""
        dev = _lib.require_gpu()
        assert precision in (None, "default", "parity"), precision
        if state_dict is None:
            state_dict = ClipWrapper._load_checkpoint(clip_model_type)
        ClipWrapper.device = dev
        ClipWrapper.clip_model_type = clip_model_type
        ClipWrapper.engine = make_vision_engine(state_dict, chunk_tiles=chunk_tiles, max_labels=max_labels)
        if precision == "parity":
            if not hasattr(ClipWrapper.engine, "qk_split"):
                raise NotImplementedError('precision="parity" is implemented for the ViT-B towers (closed-form rollout)')
            ClipWrapper.engine.qk_split = True
        ClipWrapper.text = TextEncoder(state_dict) if "token_embedding.weight" in state_dict else None
        ClipWrapper._coeffs = _ResizeCoeffs(dev)
        # ToTensor (/255) then Normalize, all fp32 like torchvision, then rounded once to the GEMM operand type
        u = np.arange(256, dtype=np.float32)[None, :] / np.float32(255)
        lut = (u - np.asarray(CLIP_MEAN, np.float32)[:, None]) / np.asarray(CLIP_STD, np.float32)[:, None]
        ClipWrapper._lut = torch.from_numpy(lut.astype(np.float32)).to(dev, torch.float16).contiguous()
        ClipWrapper.class_to_language_feature = {}
        ClipWrapper._patches = {}
        ClipWrapper._plans = {}
        ClipWrapper._jitter_scratch = None

    @staticmethod
    def _load_checkpoint(clip_model_type):
        if ClipWrapper.state_dict_provider is not None:
            return ClipWrapper.state_dict_provider(clip_model_type)
        fname = {"ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt"}.get(clip_model_type)
        if fname is None:
            raise RuntimeError(f"Model {clip_model_type} not found; available models = ['ViT-B/32', 'ViT-B/16', 'ViT-L/14']")
        path = clip_model_type if os.path.isfile(clip_model_type) else os.path.expanduser(f"~/.cache/clip/{fname}")
        if not os.path.isfile(path):
            raise RuntimeError(f"CLIP checkpoint {path} not found (no network here): put the OpenAI checkpoint there or "
                               "pass state_dict= / set ClipWrapper.state_dict_provider")
        try:
            return torch.jit.load(path, map_location="cpu").state_dict()
        except RuntimeError:
            return torch.load(path, map_location="cpu")

    @classmethod
    def check_initialized(cls, clip_model_type="ViT-B/32", **kwargs):
        if cls.engine is None:
            ClipWrapper(clip_model_type=clip_model_type, **kwargs)

    # ---- text ------------------------------------------------------------------------------------
    @classmethod
    def set_classes(cls, classes: Sequence[str]):
        """zeroshot_classifier (clip_gradcam.py:12-27, 134-142) for `classes` x `cls.templates`."""
        cls.check_initialized()
        if cls.tokenizer is None:
            from .tokenizer import BPETokenizer
            cls.tokenizer = BPETokenizer()
        classes = [str(c) for c in classes]
        texts = [t.format(c) for c in classes for t in cls.templates]
        w = cls.set_classes_tokens(classes, cls.tokenizer.tokenize(texts), len(cls.templates))
        return w

    @classmethod
    def set_classes_tokens(cls, classes: Sequence[str], tokens: torch.Tensor, n_templates: int):
        cls.check_initialized()
        if cls.text is None:
            raise RuntimeError("the loaded state dict has no text tower")
        w = cls.text.zeroshot_weights(tokens, len(classes), n_templates)            # [L, E] on the GPU
        cls.class_to_language_feature = {c: w[i] for i, c in enumerate(classes)}
        return w

    @classmethod
    def get_clip_text_feature(cls, string):
        cls.check_initialized()
        if cls.tokenizer is None:
            from .tokenizer import BPETokenizer
            cls.tokenizer = BPETokenizer()
        strings = [string] if isinstance(string, str) else list(string)
        tok = cls.tokenizer.tokenize(strings)
        # encode_text without the zero-shot normalisation: one "template" per string, raw projection
        e = cls.text.encode(tok)
        return e.squeeze().cpu().numpy()

    # ---- relevancy -----------------------------------------------------------------------------------
    @classmethod
    def get_clip_saliency(cls, img, text_labels, prompts, distractor_labels=set(), use_lavt=False, **kwargs):
        cls.check_initialized()
        if use_lavt:
            raise NotImplementedError("LAVT is not part of the reference either (ClipWrapper.lavt is never set)")
        cls.templates = prompts
        text_labels = [str(t) for t in text_labels]
        w = cls.set_classes(text_labels)
        text_label_features = w.cpu()
        text_maps = cls.get_clip_saliency_convolve(img=img, text_labels=text_labels, **kwargs)
        if len(distractor_labels) > 0:
            distractor_labels = set(distractor_labels) - set(text_labels)
            cls.set_classes(list(distractor_labels))
            distractor_maps = cls.get_clip_saliency_convolve(img=img, text_labels=list(distractor_labels), **kwargs)
            text_maps -= distractor_maps.mean(dim=0)
        return text_maps.cpu(), text_label_features

    @classmethod
    def get_clip_saliency_convolve(cls, text_labels, horizontal_flipping=False, positive_attn_only: bool = False,
                                   tile_batch_size=32, prompt_batch_size=32, tile_interpolate_batch_size=32, **kwargs):
        cls.positive_attn_only = positive_attn_only
        w = torch.stack([cls.class_to_language_feature[str(c)] for c in text_labels], dim=0).contiguous()
        images = cls.make_images(kwargs["img"], kwargs["augmentations"], kwargs.get("jittered_images"))
        return cls.relevancy_device(images, w, kwargs["cropping_augmentations"], horizontal_flipping, positive_attn_only).cpu()

    @classmethod
    def make_images(cls, img, augmentations: int, jittered_images=None, img_dev: torch.Tensor | None = None, seed: int | None = None) -> torch.Tensor:
        """uint8 [1 + augmentations, H, W, 3] on the GPU: the image then its colour-jittered copies (img_dev: the image already in HBM).
        seed: draw the jitter parameters from np.random.default_rng(seed) instead of the wrapper's running generator - ranks that share
        one scene (tile sharding) must jitter identically."""
        assert type(img) == np.ndarray and img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
        cls.check_initialized()
        dev = cls.device
        base = img_dev if img_dev is not None else torch.from_numpy(np.ascontiguousarray(img)).to(dev)
        if augmentations == 0:
            return base[None].contiguous()
        if jittered_images is not None:
            assert len(jittered_images) == augmentations
            return torch.stack([base] + [torch.from_numpy(np.ascontiguousarray(j)).to(dev) for j in jittered_images])
        H, W = img.shape[:2]
        base = base.contiguous()
        out = torch.empty(augmentations + 1, H, W, 3, dtype=torch.uint8, device=dev)
        if (H * W * 3) % 4 == 0:
            _lib.call("semabs_replicate", _lib.ptr(base), _lib.ptr(out), H * W * 3, augmentations + 1, _lib.stream())
        else:
            out.copy_(base[None].expand_as(out))
        if cls._jitter_scratch is None:
            cls._jitter_scratch = torch.empty(1, dtype=torch.int64, device=dev)      # zeroed by the contrast step itself
        scratch = cls._jitter_scratch
        rng = cls._rng if seed is None else np.random.default_rng(seed)
        for k in range(1, augmentations + 1):
            order = rng.permutation(4)
            f = [rng.uniform(0.4, 1.6), rng.uniform(0.4, 1.6), rng.uniform(0.4, 1.6), rng.uniform(-0.1, 0.1)]
            _lib.call("semabs_color_jitter", out[k].data_ptr(), H, W, _lib.iarr(order), _lib.farr(f), _lib.ptr(scratch), _lib.stream())
        return out

    @classmethod
    def relevancy_device(cls, images: torch.Tensor, w_text: torch.Tensor, cropping_augmentations, horizontal_flipping: bool,
                         positive_attn_only: bool, tile_range=None, return_tiles: bool = False):
        """images uint8 [n_img, H, W, 3] (GPU), w_text fp32 [L, E] (GPU) -> fp32 [L, H, W] on the GPU.
        tile_range=(t0, t1): only run the ViT on that slice of the tile table (multi-GPU tile sharding); with return_tiles the result is
        this rank's slice [L, t1 - t0, g, g] per pass, which the caller all-gathers (`dist.allgather_tile_relevance`) before aggregation."""
        cls.check_initialized()
        eng = cls.engine
        dev = cls.device
        n_img, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        table, scales, tiles_dev, _ = cls._plan(H, W, n_img, cropping_augmentations)
        N = len(table)
        L = int(w_text.shape[0])
        g = eng.g
        xmin_d, kk_d, ks_d = cls._coeffs.device()
        max_ks = max(cls._coeffs.ksize) if cls._coeffs.ksize else 0
        passes = 2 if horizontal_flipping else 1
        G, Kp = g * g, 3 * eng.p * eng.p
        t_lo, t_hi = (0, N) if tile_range is None else tile_range
        # relevance of every (pass, tile) forward of this call, in issue order: pass p, tile t at column p * (t_hi - t_lo) + (t - t_lo),
        # so that one ViT batch may span the flip boundary
        # (every cell is written by semabs_rollout: one launch per label group covers all (pass, tile) columns of the chunk)
        rel_all = torch.empty(L, passes * (t_hi - t_lo), g, g, dtype=torch.float32, device=dev)
        # Tile chunks are independent: alternate them over `n_streams` HIP streams (one workspace each) so one chunk's
        # memory-bound kernels and GEMM store tails overlap the other chunk's MFMA phases.
        main = torch.cuda.current_stream()
        # The (pass, tile) forwards form ONE list cut into chunks of `eng.chunk`: a single ragged batch at the very end instead of one
        # per pass (the chunk size is tuned so that full batches have no ragged last wave of GEMM tiles).
        n_per = t_hi - t_lo
        work = []                                       # per chunk: list of (flip, first tile, count)
        for v0 in range(0, passes * n_per, eng.chunk):
            v1 = min(passes * n_per, v0 + eng.chunk)
            segs = []
            for flip in range(passes):
                a, b = max(v0, flip * n_per), min(v1, (flip + 1) * n_per)
                if a < b:
                    segs.append((flip, t_lo + a - flip * n_per, b - a))
            work.append(segs)
        ns = max(1, min(cls.n_streams, len(work)))
        if cls._streams is None or len(cls._streams) < ns:
            cls._streams = cls._make_streams(ns)
        # Patch matrix of EVERY (pass, tile) forward of this call, in issue order (737 MB at the BASELINE shape): each tile is resampled
        # once and written as is and mirrored (the flip pass differs only in the output column), in one launch of 14 workgroups per tile
        # instead of one launch per chunk and pass; a chunk's GEMM A operand is a row slice of it.
        key = (passes * n_per * G, Kp)
        if cls._patches.get("key") != key:
            cls._patches = {"key": key, "buf": torch.empty(key[0], Kp, dtype=torch.float16, device=dev)}
        patches_all = cls._patches["buf"]
        if n_per > 0:
            _lib.call("semabs_tile_patches", _lib.ptr(images), n_img, H, W, tiles_dev[t_lo:].data_ptr(), n_per, _lib.ptr(xmin_d),
                      _lib.ptr(kk_d), _lib.ptr(ks_d), _lib.ptr(cls._lut), _lib.ptr(patches_all), eng.p, 2 if passes == 2 else 0, max_ks, _lib.stream())
        for i in range(ns):
            cls._streams[i].wait_stream(main)
        w_chunks = [w_text[l0:l0 + eng.max_labels].contiguous() for l0 in range(0, L, eng.max_labels)]
        for i, segs in enumerate(work):
            slot = i % ns
            with torch.cuda.stream(cls._streams[slot]):
                eng.slot = slot
                m = sum(cnt for _, _, cnt in segs)
                v0 = i * eng.chunk                              # first (pass, tile) forward of the chunk = row v0 * G of the patch matrix
                patches = patches_all[v0 * G:(v0 + m) * G]
                if isinstance(eng, VisionRollout):              # ln_pre + block 0's ln_1 as one pass when the batch takes the folded launch sequence
                    eng.embed(patches, m, defer_ln_pre=os.environ.get("SEMABS_LN2", "1") == "1")
                else:
                    eng.embed(patches, m)
                eng.trunk(m); eng.head(m)
                col0 = i * eng.chunk                            # the chunk's tiles are consecutive columns of rel_all
                for li, wl in enumerate(w_chunks):
                    l0 = li * eng.max_labels
                    eng.rollout(m, wl, positive_attn_only, rel_all[l0:l0 + eng.max_labels], col0)
        for i in range(ns):
            main.wait_stream(cls._streams[i])
        eng.slot = 0
        for t in (images, tiles_dev, w_text, rel_all, patches_all, *w_chunks):
            for i in range(ns):
                t.record_stream(cls._streams[i])
        if tile_range is None and not return_tiles:
            # both passes stay in rel_all: the un-flip average reads them in place (no slice copies) and the aggregation takes its result
            return cls.aggregate_device([rel_all], scales, n_img, H, W, passes=passes, plan_key=(H, W, n_img, cls._aug_key(cropping_augmentations)))
        if tile_range is None:
            rel = [rel_all[:, p * N:(p + 1) * N].contiguous() if passes > 1 else rel_all for p in range(passes)]
        else:                                           # a shard: only this rank's tile slice [L, t_hi - t_lo, g, g] per pass (all-gathered by the caller)
            rel = [rel_all[:, p * n_per:(p + 1) * n_per].contiguous() for p in range(passes)]
        if return_tiles:
            return rel, table, scales
        return cls.aggregate_device(rel, scales, n_img, H, W)

    @classmethod
    def _make_streams(cls, ns: int):
        return [torch.cuda.Stream() for _ in range(ns)]

    @staticmethod
    def _aug_key(cropping_augmentations):
        return tuple((int(a["tile_size"]), int(a["stride"])) for a in cropping_augmentations)

    @classmethod
    def _plan(cls, H: int, W: int, n_img: int, cropping_augmentations):
        """Tile table, scale table and their device copies for one (image shape, image count, crop configuration): built and uploaded ONCE - a
        scene of a stream of equally-shaped frames issues no host -> device copy for them."""
        key = (H, W, n_img, cls._aug_key(cropping_augmentations))
        pl = cls._plans.get(key)
        if pl is None or pl[4] != len(cls._coeffs.sizes):
            table, scales = plan_tiles(H, W, n_img, cropping_augmentations)
            coef_ids = np.asarray([cls._coeffs.id_of(int(ts)) for ts in table[:, 3]], np.int32)
            tiles_dev = torch.from_numpy(np.concatenate([table, coef_ids[:, None]], axis=1).astype(np.int32)).to(cls.device).contiguous()
            scales_dev = torch.from_numpy(np.ascontiguousarray(scales, np.int32)).to(cls.device)
            if len(cls._plans) > 64:
                cls._plans.clear()
            pl = cls._plans[key] = (table, scales, tiles_dev, scales_dev, len(cls._coeffs.sizes))
        return pl[0], pl[1], pl[2], pl[3]

    @classmethod
    def aggregate_device(cls, rel, scales: np.ndarray, n_img: int, H: int, W: int, passes: int | None = None, plan_key=None) -> torch.Tensor:
        """rel: [pass-0 maps, pass-1 maps] (each [L, N, g, g]), or ONE buffer [L, passes * N, g, g] holding both passes with `passes` given."""
        dev = cls.device
        fused = passes is not None
        L, g = int(rel[0].shape[0]), int(rel[0].shape[2])
        N = int(rel[0].shape[1]) // (passes if fused else 1)
        out = torch.empty(L, H, W, dtype=torch.float32, device=dev)
        pl = cls._plans.get(plan_key) if plan_key is not None else None
        sc = pl[3] if pl is not None else torch.from_numpy(np.ascontiguousarray(scales, np.int32)).to(dev)
        maps = rel[0]
        if fused and passes > 1:
            maps = torch.empty(L, N, g, g, dtype=torch.float32, device=dev)
            _lib.call("semabs_unflip_average_rows", _lib.ptr(rel[0]), rel[0].data_ptr() + N * g * g * 4, _lib.ptr(maps), L, N, passes * N, g, _lib.stream())
        elif len(rel) > 1:                              # un-flip average once per map cell instead of once per covered pixel (bit-identical, half the loads)
            maps = torch.empty_like(rel[0])
            _lib.call("semabs_unflip_average", _lib.ptr(rel[0]), _lib.ptr(rel[1]), _lib.ptr(maps), L * N, g, _lib.stream())
        _lib.call("semabs_aggregate", _lib.ptr(maps), None, L, N, g, H, W, _lib.ptr(sc), len(scales), n_img, N // n_img, _lib.ptr(out), _lib.stream())
        return out


__all__ = ["ClipWrapper", "saliency_configs", "imagenet_templates", "plan_tiles"]
