"""Drop-in for the reference's `unet3d.ResidualUNet3D` (unet3d.py:658-689 -> Abstract3DUNet :481-621), inference only,
computed by the HIP implicit-GEMM kernels in csrc/unet.hip.

Same constructor arguments, same `state_dict` key names (`encoders.{i}.basic_module.conv{1,2,3}.{groupnorm,conv}.*`,
`decoders.{i}.upsampling.upsample.*`, `final_conv.*`), `forward(x[N, C, D, H, W]) -> [N, out, D, H, W]`.
Internally activations are channels-last [N, D, H, W, C]; `forward_cl` takes / returns that layout directly
(what `SemAbs3D` uses, so nothing is transposed on the hot path).

precision = "fp16": fp16 activations in HBM, one fp16 MFMA per k-step, fp32 accumulate, fp64 GroupNorm statistics.
precision = "exact": fp32 activations, operands split into fp16 hi + lo (3 MFMAs) - ~fp32 accuracy, used to show
that the residual L-inf of the fp16 mode is rounding, not a defect.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np
import torch

from . import _lib


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]


def _pack_fragments(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Kp] (Kp % 32 == 0, Cout % 16 == 0) -> [Kp / 32][Cout / 16][lane = kg * 16 + row][8] flattened, k = 32 * k-step + 8 * kg + e:
    the order in which a wave's 64 lanes hold the A operand of v_mfma_f32_16x16x32_f16 (csrc/unet.hip, SEMABS_CONV_PACKED)."""
    cout, kp = int(w.shape[0]), int(w.shape[1])
    assert cout % 16 == 0 and kp % 32 == 0
    # w[cb * 16 + vl, ks * 32 + kg * 8 + e] -> [ks, cb, kg, vl, e]
    return w.reshape(cout // 16, 16, kp // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().reshape(-1)


def _split16(w: torch.Tensor, dev):
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return hi.to(dev).contiguous(), lo.to(dev).contiguous()


class _Conv:
    """GroupNorm (optional) + Conv3d k^3 pad k//2, weights re-laid out as [Cout, (kd, kh, kw, cin)] padded to 32."""

    def __init__(self, weight: torch.Tensor, gn_w, gn_b, bias, groups: int, dev):
        cout, cin, k = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        w = weight.float().permute(0, 2, 3, 4, 1).reshape(cout, k * k * k * cin)
        kp = (w.shape[1] + 31) // 32 * 32
        if kp != w.shape[1]:
            w = torch.cat([w, torch.zeros(cout, kp - w.shape[1], device=w.device)], dim=1)
        # Fragment-packed copy appended behind the [Cout, Kp] matrix (flag bit 9 of the convolution entry points' flag word,
        # SEMABS_CONV_PACKED): one MFMA A operand = 1 KB of contiguous memory (see _pack_fragments)
        self.packed = 0
        flat = w.contiguous().reshape(-1)
        if cout % 16 == 0:
            flat = torch.cat([flat, _pack_fragments(w)])
            self.packed = 512
        self.w_hi, self.w_lo = _split16(flat, dev)
        self.cin, self.cout, self.k = cin, cout, k
        self.gn_w = None if gn_w is None else gn_w.float().to(dev).contiguous()
        self.gn_b = None if gn_b is None else gn_b.float().to(dev).contiguous()
        self.bias = None if bias is None else bias.float().to(dev).contiguous()
        self.groups = groups if cin >= groups else 1


class _ConvT:
    """ConvTranspose3d k3 s2 p1 (+ output_padding 1 via output_size): 8 parity-class weight matrices."""

    def __init__(self, weight: torch.Tensor, bias, dev):
        cin, cout = int(weight.shape[0]), int(weight.shape[1])
        w = weight.float()
        mats, packs, offs, off = [], [], [], 0
        for cls in range(8):
            p = (cls >> 2, (cls >> 1) & 1, cls & 1)
            cols = []
            for t0 in range(p[0] + 1):
                for t1 in range(p[1] + 1):
                    for t2 in range(p[2] + 1):
                        k = [1 if pp == 0 else (0 if t == 0 else 2) for pp, t in zip(p, (t0, t1, t2))]
                        cols.append(w[:, :, k[0], k[1], k[2]].t())          # [Cout, Cin]
            m = torch.cat(cols, dim=1).contiguous()                            # [Cout, ntaps * Cin]
            mats.append(m.reshape(-1))
            packs.append(_pack_fragments(m) if cout % 16 == 0 and cin % 32 == 0 else None)
            offs.append(off)
            off += m.numel()
        self.packed = 512 if all(p is not None for p in packs) else 0       # fragment-packed copies of the class matrices behind the plain ones
        flat = torch.cat(mats + (packs if self.packed else []))
        self.w_hi, self.w_lo = _split16(flat, dev)
        self.class_off = (C.c_long * 8)(*offs)
        self.bias = bias.float().to(dev).contiguous()
        self.cin, self.cout = cin, cout


class ResidualUNet3D(torch.nn.Module):
    """nn.Module surface of the reference (`parameters()`, `state_dict()` / `load_state_dict()`, `.to()`, `train()` / `eval()`); the forward pass
    is inference-only (no autograd graph) and runs on the HIP kernels with operands derived from the parameters (see module.py)."""

    def __init__(self, in_channels, out_channels, f_maps=64, num_groups=8, num_levels=5, final_sigmoid=False,
                 layer_order="gcr", is_segmentation=False, precision: str = "exact", **kwargs):
        super().__init__()
        assert layer_order == "gcr" and not is_segmentation, "the path uses order 'gcr' without a final activation"
        assert precision in ("fp16", "exact")
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels=num_levels)
        self.f_maps = list(f_maps)
        self.in_channels, self.out_channels, self.num_groups = in_channels, out_channels, num_groups
        # "exact" (fp32 activations, split-fp16 MFMA operands: the reference's fp32 results to ~1e-5) is the default of the reference-surface
        # classes; "fp16" is the opt-in fast mode (activations rounded to fp16: 3.6e-3 on the voxel logits, measured)
        self.precision = precision
        self.f32 = int(precision == "exact")
        self.act_dtype = torch.float32 if self.f32 else torch.float16
        from .module import register_tree
        from .weights import make_unet_state_dict
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())            # follows torch's RNG like the reference's default init (seed_all)
        register_tree(self, make_unet_state_dict(seed, in_channels, out_channels, self.f_maps[0], len(self.f_maps)))
        self._sums_arena = None
        self.enc: List[List[_Conv]] = []
        self.dec: List = []
        self.final = None
        self._sig = None

    @property
    def dev(self):
        return _lib.require_gpu()

    # ---- weights -----------------------------------------------------------------------------------
    def expected_keys(self):
        from .weights import unet_layer_plan
        keys = []
        for pre, kind, cin, cout in unet_layer_plan(self.in_channels, self.out_channels, self.f_maps[0], len(self.f_maps)):
            if kind in ("gcr", "gc"):
                keys += [pre + "groupnorm.weight", pre + "groupnorm.bias", pre + "conv.weight"]
            else:
                keys += [pre + "weight", pre + "bias"]
        return keys

    def load_state_dict(self, state_dict, strict: bool = True, prefix: str = "", **kw):
        """nn.Module.load_state_dict (missing / unexpected keys raise under strict=True, are reported under strict=False) + an optional key
        `prefix` to pick this module's entries out of an enclosing state dict."""
        from .module import strip_module_prefix
        sd = strip_module_prefix(state_dict)
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        return super().load_state_dict(sd, strict=strict, **kw)

    def _sync(self):
        """(Re)build the kernel operands when a parameter changed (module.signature)."""
        from .module import signature
        dev = _lib.require_gpu()
        if any(p.device.type != "cuda" for p in self.parameters()):
            self.to(dev)                                # there is no CPU path: a module left on the host is moved to the HIP device on first use
        sig = signature(self)
        if sig == self._sig:
            return
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        G = self.num_groups

        def block(pre):
            return [_Conv(sd[pre + f"conv{j}.conv.weight"], sd[pre + f"conv{j}.groupnorm.weight"],
                          sd[pre + f"conv{j}.groupnorm.bias"], None, G, dev) for j in (1, 2, 3)]

        L = len(self.f_maps)
        self.enc = [block(f"encoders.{i}.basic_module.") for i in range(L)]
        self.dec = [(_ConvT(sd[f"decoders.{i}.upsampling.upsample.weight"], sd[f"decoders.{i}.upsampling.upsample.bias"], dev),
                     block(f"decoders.{i}.basic_module.")) for i in range(L - 1)]
        self.final = _Conv(sd["final_conv.weight"], None, None, sd["final_conv.bias"], G, dev)
        self._sig = sig

    # ---- kernels -----------------------------------------------------------------------------------
    def _zero_sums(self, B: int, G: int) -> torch.Tensor:
        """fp64 [B, G, 2] of zeros for one layer's GroupNorm statistics, cut from an arena that ONE fill zeroes per forward pass (a forward needs ~30
        of them: one launch instead of thirty).  Outside a forward pass (no arena open) it is a plain zero-filled buffer."""
        n = B * G * 2
        a = self._sums_arena
        if a is None or a[1] + n > a[0].numel():
            return _lib.filled((B, G, 2), torch.float64, 0, self.dev)
        t = a[0][a[1]:a[1] + n].view(B, G, 2)
        a[1] += n
        return t

    def _gn(self, x, conv: _Conv, sums=None):
        """GroupNorm scale / shift [B, C] of `conv`'s input x; `sums` = statistics already produced by the kernel that wrote x."""
        B, nvox, Cc = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3], x.shape[4]
        G = conv.groups
        st = _lib.stream()
        if sums is None:
            sums = self._zero_sums(B, G)
            _lib.call("semabs_gn_stats", _lib.ptr(x), _lib.ptr(sums), B, nvox, Cc, G, self.f32, st)
        scale = torch.empty(B, Cc, dtype=torch.float32, device=self.dev)
        shift = torch.empty(B, Cc, dtype=torch.float32, device=self.dev)
        _lib.call("semabs_gn_finalize", _lib.ptr(sums), _lib.ptr(conv.gn_w), _lib.ptr(conv.gn_b), _lib.ptr(scale), _lib.ptr(shift),
                  B, Cc, G, nvox, 1e-5, st)
        return scale, shift

    def _conv(self, x, conv: _Conv, relu, resid=None, gn=True, out_dtype=None, in_sums=None, out_groups=0, generic=False, occ=None):
        """out_groups > 0: also return the GroupNorm statistics of the output (fp64 [B, out_groups, 2]) for the next layer.
        generic: run the generic gather kernel even where an LDS-brick kernel exists (bit 8 of the flag word; per-call cross-check for tests).
        occ: x is SPARSE (semabs_scatter_mean_sparse): uint32 occupancy bitmap; voxels whose bit is clear are zero and are not read."""
        B, D0, D1, D2, _ = x.shape
        y = torch.empty(B, D0, D1, D2, conv.cout, dtype=self.act_dtype, device=self.dev)
        scale, shift = self._gn(x, conv, in_sums) if gn else (None, None)
        args = (_lib.ptr(x), _lib.ptr(conv.w_hi), _lib.ptr(conv.w_lo), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift),
                _lib.ptr(conv.bias), _lib.ptr(resid), B, D0, D1, D2, conv.cin, conv.cout, conv.k, int(relu), self.f32 | (256 if generic else 0) | conv.packed)
        if out_groups:
            sums = self._zero_sums(B, out_groups)
            if occ is not None:
                _lib.call("semabs_conv3d_sparse_stats", args[0], _lib.ptr(occ), *args[1:], _lib.ptr(sums), out_groups, _lib.stream())
            else:
                _lib.call("semabs_conv3d_stats", *args, _lib.ptr(sums), out_groups, _lib.stream())
            return y, sums
        assert occ is None
        _lib.call("semabs_conv3d", *args, _lib.stream())
        return y

    @staticmethod
    def sparse_input_supported(shape, conv: "_Conv") -> bool:
        """The first convolution can read a sparse (bitmap-described) input: the 16 -> 16 level-0 kernel's shapes."""
        _, D0, D1, D2, Cc = shape
        return conv.k == 3 and conv.cin == 16 and conv.cout == 16 and Cc == 16 and D0 % 8 == 0 and D1 % 8 == 0 and D2 % 16 == 0 and D0 * D1 * D2 * 16 < 2 ** 31

    def _block(self, x, convs, in_sums=None, occ=None):
        # conv1 / conv2 hand the statistics of their outputs to the GroupNorm of conv2 / conv3 (fused into the epilogue where supported);
        # in_sums: the statistics of x when its producer already has them (the transposed convolution of a decoder level)
        out1, s1 = self._conv(x, convs[0], relu=True, in_sums=in_sums, out_groups=convs[1].groups, occ=occ)
        out2, s2 = self._conv(out1, convs[1], relu=True, in_sums=s1, out_groups=convs[2].groups)
        return self._conv(out2, convs[2], relu=True, resid=out1, in_sums=s2)          # conv3 (no ReLU) + residual, then ReLU

    def _pool(self, x):
        B, D0, D1, D2, Cc = x.shape
        y = torch.empty(B, D0 // 2, D1 // 2, D2 // 2, Cc, dtype=self.act_dtype, device=self.dev)
        _lib.call("semabs_maxpool3d", _lib.ptr(x), _lib.ptr(y), B, D0, D1, D2, Cc, self.f32, _lib.stream())
        return y

    def _up(self, x, skip, ct: _ConvT, out_groups=0, generic=False):
        """ConvTranspose3d(k3, s2) + skip; out_groups > 0: also the GroupNorm statistics of the result for the block that follows."""
        B, D0, D1, D2, _ = x.shape
        assert tuple(skip.shape) == (B, 2 * D0, 2 * D1, 2 * D2, ct.cout)
        y = torch.empty_like(skip)
        args = (_lib.ptr(x), _lib.ptr(ct.w_hi), _lib.ptr(ct.w_lo), ct.class_off, _lib.ptr(y),
                _lib.ptr(ct.bias), _lib.ptr(skip), B, D0, D1, D2, ct.cin, ct.cout, self.f32 | (256 if generic else 0) | ct.packed)
        if out_groups:
            sums = self._zero_sums(B, out_groups)
            _lib.call("semabs_convtranspose3d_stats", *args, _lib.ptr(sums), out_groups, _lib.stream())
            return y, sums
        _lib.call("semabs_convtranspose3d", *args, _lib.stream())
        return y

    # ---- forward -----------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_cl(self, x: torch.Tensor, taps: dict | None = None, skip_final: bool = False, in_sums=None, occ=None) -> torch.Tensor:
        """x [B, D0, D1, D2, Cin] channels-last (act dtype, GPU) -> [B, D0, D1, D2, Cout]  (skip_final: the input of `final_conv`;
        in_sums: GroupNorm statistics of x, fp64 [B, groups, 2], when its producer already has them; occ: x is sparse - occupancy bitmap, see _conv)."""
        self._sync()
        assert x.dtype == self.act_dtype and x.is_contiguous()
        # statistics arena of this pass: (3 per residual block + 1 per transposed convolution) x [B, <= num_groups, 2] doubles, zeroed by one launch;
        # a fresh allocation per pass (the caching allocator recycles it), so passes on different streams never share one
        n_layers = 3 * (len(self.enc) + len(self.dec)) + len(self.dec) + 2
        self._sums_arena = [_lib.filled((n_layers * int(x.shape[0]) * max(1, self.num_groups) * 2,), torch.float64, 0, self.dev), 0]
        try:
            return self._forward_cl(x, taps, skip_final, in_sums, occ)
        finally:
            self._sums_arena = None

    def _forward_cl(self, x, taps, skip_final, in_sums, occ=None):
        feats = []
        for i, convs in enumerate(self.enc):
            if i > 0:
                x = self._pool(x)
            x = self._block(x, convs, in_sums=in_sums if i == 0 else None, occ=occ if i == 0 else None)
            if taps is not None:
                taps[f"enc{i}"] = x
            feats.insert(0, x)
        for i, (skip, (ct, convs)) in enumerate(zip(feats[1:], self.dec)):
            x, sums = self._up(x, skip, ct, out_groups=convs[0].groups)
            x = self._block(x, convs, in_sums=sums)
            if taps is not None:
                taps[f"dec{i}"] = x
        if skip_final:
            return x
        return self._conv(x, self.final, relu=False, gn=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """NCDHW fp32 in / out like the reference (two layout copies around the channels-last core)."""
        dev = _lib.require_gpu()
        xc = x.to(dev).permute(0, 2, 3, 4, 1).contiguous().to(self.act_dtype)
        y = self.forward_cl(xc)
        return y.permute(0, 4, 1, 2, 3).contiguous().float()
