"""One optimisation step of SemAbsVOOL on the GPU (config 5 of BASELINE.json: `train_vool.py` -> `utils.loop`, utils.py:404-417):
forward (net.py:506-579), BCE-with-logits (train_vool.py:171-178), backward, clip_grad_norm_ (utils.py:415), Lamb.step
(arm/optim/lamb.py:59-127), all on the HIP kernels of csrc/train.hip + csrc/unet.hip + csrc/optim.hip.  There is no autograd graph:
the backward pass is written out layer by layer against a tape of saved activations, fp32 ("exact" mode) throughout.

`VOOLTrainer` keeps fp32 master parameters under the reference's `state_dict` key names, one flat gradient buffer (a single
RCCL all-reduce per step when torch.distributed is initialised - what DistributedDataParallel's buckets do, utils.py:283-288), and
the `Lamb` optimiser of optim.py.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .net import VirtualGrid
from .optim import Lamb
from .unet3d import number_of_features_per_level
from .weights import unet_layer_plan

RELATIONS = ["in", "behind", "in front of", "on the left of", "on the right of", "on", "[pad]"]
SLOPE = 0.01


def _taps(vals) -> bytes:
    return np.asarray(vals, np.int8).tobytes()


TAPS_CONV3 = _taps([(a - 1, b - 1, c - 1) for a in range(3) for b in range(3) for c in range(3)])
TAPS_ONE = _taps([(0, 0, 0)])


def _pad32(w: torch.Tensor) -> torch.Tensor:
    kp = (w.shape[1] + 31) // 32 * 32
    if kp == w.shape[1]:
        return w.contiguous()
    out = torch.zeros(w.shape[0], kp, dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


def _flat_packed(w: torch.Tensor):
    """[rows, Kp] -> (flat = the matrix followed by its fragment-packed copy when the shape allows, flag word)."""
    from .unet3d import _pack_fragments
    if w.shape[0] % 16 == 0 and w.shape[1] % 32 == 0:
        return torch.cat([w.reshape(-1), _pack_fragments(w)]), 512
    return w.reshape(-1), 0


class _WeightLayouts:
    """Split-fp16 kernel operands of the trainable weights, refreshed every step by index gathers (semabs_gather_split16).
    `build(t)` is the layout written with torch data-movement ops (permute / flip / cat / zero padding) - it is run ONCE on an index tensor
    (1 .. numel, zeros = padding) to obtain the int32 map, and the hi / lo buffers keep their addresses for the lifetime of the trainer.
    Round 6: `refresh_all()` at the start of a step re-gathers EVERY registered layout in one launch (semabs_gather_split16_batched: 78 launches of 5 - 60 us
    before); `get()` then only hands out the buffers.  A layout that is new, or whose source tensor moved, is gathered on the spot and joins the table."""

    def __init__(self, dev):
        self.dev = dev
        self.maps: Dict[str, tuple] = {}
        self.src_ptr: Dict[str, int] = {}
        self.fresh: set = set()          # keys gathered since the last weight update
        self._table = None               # (device job table, njobs, total blocks, keys)

    def invalidate(self):
        """The weights changed (optimiser step, load): every layout is stale."""
        self.fresh.clear()

    def refresh_all(self):
        if not self.maps or os.environ.get("SEMABS_BATCH_GATHER", "1") != "1":      # A/B: 0 = one launch per matrix, in get()
            return
        keys = tuple(self.maps)
        if self._table is None or self._table[3] != keys or any(self._table[4][k] != self.src_ptr[k] for k in keys):
            rows, blk = [], 0
            for k in keys:
                idx, hi, lo = self.maps[k][:3]
                n = idx.numel()
                rows.append([self.src_ptr[k], idx.data_ptr(), hi.data_ptr(), lo.data_ptr(), n, blk])
                blk += (n + 1023) // 1024
            self._table = (torch.tensor(rows, dtype=torch.int64).to(self.dev), len(rows), blk, keys, dict(self.src_ptr))
        tab, nj, nb = self._table[:3]
        _lib.call("semabs_gather_split16_batched", _lib.ptr(tab), nj, nb, _lib.stream())
        self.fresh = set(keys)

    def get(self, key: str, w: torch.Tensor, build):
        m = self.maps.get(key)
        if m is None or m[3] != tuple(w.shape):
            ar = torch.arange(1, w.numel() + 1, dtype=torch.int32, device=self.dev).view(w.shape)
            out = build(ar)
            flat, extra = (out if isinstance(out, tuple) else (out, None))
            idx = (flat.reshape(-1) - 1).to(torch.int32).contiguous()
            hi = torch.empty(idx.numel(), dtype=torch.float16, device=self.dev)
            lo = torch.empty_like(hi)
            m = (idx, hi, lo, tuple(w.shape), extra)
            self.maps[key] = m
            self.fresh.discard(key)
        idx, hi, lo, _, extra = m
        src = w.detach()
        assert src.is_contiguous() and src.dtype == torch.float32
        if key not in self.fresh or self.src_ptr.get(key) != src.data_ptr():
            self.src_ptr[key] = src.data_ptr()
            _lib.call("semabs_gather_split16", _lib.ptr(src), _lib.ptr(idx), idx.numel(), _lib.ptr(hi), _lib.ptr(lo), _lib.stream())
            self.fresh.add(key)
        return hi, lo, extra


class _ZeroArena:
    """One zero-filled device buffer per step for the dozens of small accumulators a step needs (GroupNorm sums, channel reductions,
    max-|x| words: ~8 per convolution layer, each a separate torch fill launch before) - ONE memset at the start of the step, sub-allocated in
    order.  Falls back to torch.zeros when exhausted."""

    def __init__(self, dev, nbytes=8 << 20):
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.off = 0

    def reset(self):
        self.buf.zero_()
        self.off = 0

    def zeros(self, shape, dtype):
        n = int(np.prod(shape)) * {torch.float64: 8, torch.float32: 4, torch.int32: 4, torch.int64: 8}[dtype]
        o = (self.off + 255) // 256 * 256
        if o + n > self.buf.numel():
            return torch.zeros(*shape, dtype=dtype, device=self.buf.device)
        self.off = o + n
        return self.buf[o:o + n].view(dtype).view(*shape)


class _Rec:
    """Tape entry of one GroupNorm + Conv3d: what the backward pass needs."""
    __slots__ = ("name", "x", "y", "mean", "rstd", "scale", "shift")


class UNetTrainer:
    """Forward-with-tape and backward of ResidualUNet3D (unet3d.py:190-259, 596-621), channels-last fp32."""

    def __init__(self, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], prefix: str, in_channels: int, out_channels: int,
                 f_maps, num_groups: int, num_levels: int):
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels)
        self.f_maps = list(f_maps)
        self.p, self.g, self.prefix, self.G = params, grads, prefix, num_groups
        self.plan = unet_layer_plan(in_channels, out_channels, self.f_maps[0], len(self.f_maps))
        self.mats: Dict[str, dict] = {}
        self.dev = _lib.require_gpu()
        self._wgs = None
        self._wg_routes = {}
        self.rows_linear = os.environ.get("SEMABS_ROWS_LINEAR", "1") == "1"      # 1 x 1 x 1 convolutions / MLP layers on semabs_linear_rows
        self.wgrad_tr = os.environ.get("SEMABS_WGRAD_TR", "1") == "1"      # A/B: 0 = the round-2 brick kernel (transposes while staging, atomics)
        # GroupNorm-backward reductions (sum dXn, sum dXn xhat) from the weight-gradient pass instead of a pass over (dXn, x): semabs_wgrad_conv3_gn.
        self.wgrad_gn = os.environ.get("SEMABS_WGRAD_GN", "1") == "1"      # A/B: 0 = semabs_wgrad_conv3 + semabs_chan_reduce
        self._wg_gn_ok = {}
        # ... and, the sums being known before the data gradient, the GroupNorm-backward apply as that convolution's epilogue: semabs_conv3d_gnbwd
        self.fuse_gn_apply = os.environ.get("SEMABS_FUSE_GN_APPLY", "1") == "1"      # A/B: 0 = semabs_conv3d + semabs_gn_bwd_apply
        self._gnbwd_ok = {}
        self.mfma_wgrad = True          # tests / tuning: False = the fp32 VALU reduction kernel for every conv weight gradient
        self.debug = None               # tests: list collecting (tape kind, incoming gradient) during backward
        self.arena = _ZeroArena(self.dev)
        self.layouts = _WeightLayouts(self.dev)

    # ---- per-step weight layouts (fp16 hi/lo splits for the MFMA kernels) --------------------------------------------------------
    def refresh(self):
        """Once per step, BEFORE the forward pass: the weights moved since the last call (optimiser step / load), so every layout is re-gathered - all the
        layouts registered so far in one launch; first-time layouts join one by one in `get`."""
        L = self.layouts
        L.invalidate()
        L.refresh_all()
        for pre, kind, cin, cout in self.plan:
            key = self.prefix + pre
            if kind in ("gcr", "gc"):
                w = self.p[key + "conv.weight"]
                k = w.shape[2]
                fwd = L.get(key + "fwd", w, lambda t, cout=cout: _flat_packed(_pad32(t.permute(0, 2, 3, 4, 1).reshape(cout, -1))))
                bwd = L.get(key + "bwd", w, lambda t, cin=cin: _flat_packed(_pad32(t.flip(2, 3, 4).permute(1, 2, 3, 4, 0).reshape(cin, -1))))
                self.mats[pre] = dict(kind="conv", cin=cin, cout=cout, k=k, groups=self.G if cin >= self.G else 1, fwd=fwd, bwd=bwd)
            elif kind == "convT":
                w = self.p[key + "weight"]                                             # [cin, cout, 3, 3, 3]

                def classes(t):
                    from .unet3d import _pack_fragments
                    mats, packs, offs, off = [], [], [], 0
                    for cls in range(8):
                        pp = (cls >> 2, (cls >> 1) & 1, cls & 1)
                        cols = []
                        for t0 in range(pp[0] + 1):
                            for t1 in range(pp[1] + 1):
                                for t2 in range(pp[2] + 1):
                                    kk = [1 if q == 0 else (0 if tt == 0 else 2) for q, tt in zip(pp, (t0, t1, t2))]
                                    cols.append(t[:, :, kk[0], kk[1], kk[2]].t())
                        m = torch.cat(cols, dim=1).contiguous()
                        mats.append(m.reshape(-1)); packs.append(_pack_fragments(m)); offs.append(off); off += m.numel()
                    return torch.cat(mats + packs), offs

                hi, lo, offs = L.get(key + "fwd", w, classes)
                bwd = L.get(key + "bwd", w, lambda t, cin=cin: _flat_packed(_pad32(t.permute(0, 2, 3, 4, 1).reshape(cin, -1))))   # [cin, (k, cout)]
                self.mats[pre] = dict(kind="convT", cin=cin, cout=cout, fwd=(hi, lo, 512), class_off=(C.c_long * 8)(*offs), bwd=bwd)
            else:                                                                      # final 1x1x1 conv with bias
                w = self.p[key + "weight"]
                fwd = L.get(key + "fwd", w, lambda t, cin=cin, cout=cout: _flat_packed(_pad32(t.reshape(cout, cin))))
                bwd = L.get(key + "bwd", w, lambda t, cin=cin, cout=cout: _flat_packed(_pad32(t.reshape(cout, cin).t().contiguous())))
                self.mats[pre] = dict(kind="final", cin=cin, cout=cout, k=1, fwd=fwd, bwd=bwd)

    # ---- forward -----------------------------------------------------------------------------------------------------------------
    def _conv_fwd(self, x, pre, relu, resid=None, in_sums=None, out_groups=0):
        """-> (tape record, GroupNorm statistics of the output or None).  in_sums: statistics of x when its producer already has them
        (scatter / the previous convolution / the transposed convolution: fused into their epilogues, no pass over x);
        out_groups > 0: have this convolution produce the statistics of ITS output for the layer that follows."""
        m = self.mats[pre]
        key = self.prefix + pre
        B, D0, D1, D2, Cc = x.shape
        nvox, G, st = D0 * D1 * D2, m["groups"], _lib.stream()
        r = _Rec()
        r.name, r.x = pre, x
        sums = in_sums
        if sums is None:
            sums = self.arena.zeros((B, G, 2), torch.float64)
            _lib.call("semabs_gn_stats", _lib.ptr(x), _lib.ptr(sums), B, nvox, Cc, G, 1, st)
        r.scale = torch.empty(B, Cc, dtype=torch.float32, device=self.dev)
        r.shift = torch.empty_like(r.scale)
        r.mean = torch.empty(B, G, dtype=torch.float32, device=self.dev)
        r.rstd = torch.empty_like(r.mean)
        _lib.call("semabs_gn_finalize", _lib.ptr(sums), _lib.ptr(self.p[key + "groupnorm.weight"]), _lib.ptr(self.p[key + "groupnorm.bias"]),
                  _lib.ptr(r.scale), _lib.ptr(r.shift), B, Cc, G, nvox, 1e-5, st)
        _lib.call("semabs_gn_meanrstd", _lib.ptr(sums), _lib.ptr(r.mean), _lib.ptr(r.rstd), B, G, nvox * (Cc // G), 1e-5, st)
        r.y = torch.empty(B, D0, D1, D2, m["cout"], dtype=torch.float32, device=self.dev)
        args = (_lib.ptr(x), _lib.ptr(m["fwd"][0]), _lib.ptr(m["fwd"][1]), _lib.ptr(r.y), _lib.ptr(r.scale), _lib.ptr(r.shift),
                None, _lib.ptr(resid), B, D0, D1, D2, m["cin"], m["cout"], 3, int(relu), 1 | m["fwd"][2])
        out_sums = None
        if out_groups:
            out_sums = self.arena.zeros((B, out_groups, 2), torch.float64)
            _lib.call("semabs_conv3d_stats", *args, _lib.ptr(out_sums), out_groups, st)
        else:
            _lib.call("semabs_conv3d", *args, st)
        return r, out_sums

    def _block_fwd(self, x, pre, tape, in_sums=None):
        # conv1 / conv2 hand the statistics of their outputs to the GroupNorm of conv2 / conv3 (like unet3d.ResidualUNet3D._block)
        r1, s1 = self._conv_fwd(x, pre + "conv1.", True, in_sums=in_sums, out_groups=self.mats[pre + "conv2."]["groups"])
        r2, s2 = self._conv_fwd(r1.y, pre + "conv2.", True, in_sums=s1, out_groups=self.mats[pre + "conv3."]["groups"])
        r3, _ = self._conv_fwd(r2.y, pre + "conv3.", True, resid=r1.y, in_sums=s2)
        tape.append(("block", r1, r2, r3))
        return r3.y

    def forward(self, x: torch.Tensor, in_sums=None):
        """x fp32 [B, S, S, S, Cin] -> (y [B, S, S, S, Cout], tape).  in_sums: GroupNorm statistics of x (fp64 [B, groups, 2]) when the
        producer of x already has them (semabs_scatter_mean_stats)."""
        assert x.dtype == torch.float32 and x.is_contiguous()
        st = _lib.stream()
        tape: List = []
        L = len(self.f_maps)
        feats = []
        for i in range(L):
            if i > 0:
                B, D0, D1, D2, Cc = x.shape
                y = torch.empty(B, D0 // 2, D1 // 2, D2 // 2, Cc, dtype=torch.float32, device=self.dev)
                _lib.call("semabs_maxpool3d", _lib.ptr(x), _lib.ptr(y), B, D0, D1, D2, Cc, 1, st)
                tape.append(("pool", x, i - 1))
                x = y
            x = self._block_fwd(x, f"encoders.{i}.basic_module.", tape, in_sums=in_sums if i == 0 else None)
            feats.insert(0, x)
        for i, skip in enumerate(feats[1:]):
            pre = f"decoders.{i}.upsampling.upsample."
            m = self.mats[pre]
            B, D0, D1, D2, _ = x.shape
            y = torch.empty_like(skip)
            blk = f"decoders.{i}.basic_module."
            og = self.mats[blk + "conv1."]["groups"]
            sums = self.arena.zeros((B, og, 2), torch.float64)
            _lib.call("semabs_convtranspose3d_stats", _lib.ptr(x), _lib.ptr(m["fwd"][0]), _lib.ptr(m["fwd"][1]), m["class_off"], _lib.ptr(y),
                      _lib.ptr(self.p[self.prefix + pre + "bias"]), _lib.ptr(skip), B, D0, D1, D2, m["cin"], m["cout"], 1 | m["fwd"][2],
                      _lib.ptr(sums), og, st)
            tape.append(("up", pre, x, L - 2 - i))
            x = self._block_fwd(y, blk, tape, in_sums=sums)
        m = self.mats["final_conv."]
        B, D0, D1, D2, _ = x.shape
        y = torch.empty(B, D0, D1, D2, m["cout"], dtype=torch.float32, device=self.dev)
        wf = self.p[self.prefix + "final_conv.weight"].detach()
        if self.rows_linear and m["cin"] % 4 == 0 and m["cout"] <= 128 and wf.is_contiguous():
            # the 1 x 1 x 1 convolution IS a row-linear layer over the voxels: semabs_linear_rows streams it at HBM speed (the gather kernel: 1.28 ms at 8 x 128^3)
            _lib.call("semabs_linear_rows", _lib.ptr(x), m["cin"], _lib.ptr(wf), m["cin"], 1, _lib.ptr(self.p[self.prefix + "final_conv.bias"]), _lib.ptr(y),
                      B * D0 * D1 * D2, m["cin"], m["cout"], 0, 0.0, None, None, None, None, None, None, st)
        else:
            _lib.call("semabs_conv3d", _lib.ptr(x), _lib.ptr(m["fwd"][0]), _lib.ptr(m["fwd"][1]), _lib.ptr(y), None, None,
                      _lib.ptr(self.p[self.prefix + "final_conv.bias"]), None, B, D0, D1, D2, m["cin"], m["cout"], 1, 0, 1 | m["fwd"][2], st)
        tape.append(("final", x))
        return y, tape

    # ---- backward ----------------------------------------------------------------------------------------------------------------
    def _colsum(self, a2d: torch.Tensor, grad: torch.Tensor):
        R, Cc = a2d.shape
        red = self.arena.zeros((1, Cc, 2), torch.float64)
        _lib.call("semabs_chan_reduce", _lib.ptr(a2d), None, None, None, _lib.ptr(red), 1, R, Cc, 1, _lib.stream())
        grad.add_(red[0, :, 0].float())

    def _wgrad_conv3_route(self, D0, D1, D2, ca, cx, scratch_floats) -> int:
        key = (D0, D1, D2, ca, cx, scratch_floats)
        r = self._wg_routes.get(key)
        if r is None:
            import ctypes as C
            k = C.c_int(0)
            _lib.call("semabs_wgrad_conv3_supported", D0, D1, D2, ca, cx, int(scratch_floats), C.byref(k))
            r = self._wg_routes[key] = int(k.value)
        return r

    def _wgrad_conv3_gn_ok(self, B, D0, D1, D2, ca, cx, scratch_floats) -> bool:
        key = (B, D0, D1, D2, ca, cx, scratch_floats)
        r = self._wg_gn_ok.get(key)
        if r is None:
            import ctypes as C
            k = C.c_int(0)
            _lib.call("semabs_wgrad_conv3_gn_supported", B, D0, D1, D2, ca, cx, int(scratch_floats), C.byref(k))
            r = self._wg_gn_ok[key] = bool(k.value)
        return r

    def _conv3d_gnbwd_ok(self, B, D0, D1, D2, cin_conv, cout_conv, G, have_add) -> bool:
        key = (B, D0, D1, D2, cin_conv, cout_conv, G, have_add)
        r = self._gnbwd_ok.get(key)
        if r is None:
            import ctypes as C
            k = C.c_int(0)
            _lib.call("semabs_conv3d_gnbwd_supported", B, D0, D1, D2, cin_conv, cout_conv, G, 1 if have_add else 0, C.byref(k))
            r = self._gnbwd_ok[key] = bool(k.value)
        return r

    def _wg_scratch(self):
        """(pointer, capacity in floats) of the buffer semabs_wgrad_mfma parks its row chunks' partial sums in (64 MB, allocated once)."""
        if self._wgs is None:
            self._wgs = torch.empty(16 << 20, dtype=torch.float32, device=self.dev)
        return _lib.ptr(self._wgs), self._wgs.numel()

    def _scale(self, dz: torch.Tensor, B: int, Cc: int):
        """Dynamic power-of-two scale of a gradient tensor (csrc/train.hip, semabs_grad_scale): -> (scale_arr, shift_arr, s2).
        When dz came out of `_ew(..., want_max=True)` its max |.| is already known and the tensor is not read again."""
        sc = torch.empty(B * Cc, dtype=torch.float32, device=self.dev)
        sh = torch.empty_like(sc)
        s2 = torch.empty(2, dtype=torch.float32, device=self.dev)
        bits = getattr(dz, "_semabs_absmax", None)
        have = 1
        if bits is None:
            bits, have = self.arena.zeros((1,), torch.int32), 2
        _lib.call("semabs_grad_scale", _lib.ptr(dz), dz.numel(), _lib.ptr(sc), _lib.ptr(sh), B * Cc, _lib.ptr(s2), _lib.ptr(bits), have, _lib.stream())
        if have == 2:
            dz._semabs_absmax = bits                         # a second consumer of the same tensor (its weight AND data gradient) does not read it again
        return sc, sh, s2

    def _unscale_by(self, a, inv):
        out = torch.empty_like(a)
        _lib.call("semabs_ew", _lib.ptr(a), _lib.ptr(inv), _lib.ptr(out), a.numel(), 3, 0.0, None, _lib.stream())
        return out

    def _unscale(self, a, s2):
        out = torch.empty_like(a)
        inv = s2[1:]
        _lib.call("semabs_ew", _lib.ptr(a), _lib.ptr(inv), _lib.ptr(out), a.numel(), 3, 0.0, None, _lib.stream())
        return out

    def _ew(self, a, b, mode, want_max=False, in_scale=None):
        """in_scale (device scalar): a still carries a producer's dynamic gradient scale; multiply by in_scale[0] on the way in (saves the
        separate un-scaling pass over a)."""
        out = torch.empty_like(a)
        bits = self.arena.zeros((1,), torch.int32) if want_max else None
        if in_scale is not None:
            _lib.call("semabs_ew_scaled", _lib.ptr(a), _lib.ptr(b), _lib.ptr(in_scale), _lib.ptr(out), a.numel(), mode, SLOPE, _lib.ptr(bits), _lib.stream())
        else:
            _lib.call("semabs_ew", _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), mode, SLOPE, _lib.ptr(bits), _lib.stream())
        if want_max:
            out._semabs_absmax = bits
        return out

    def _conv_bwd(self, r: _Rec, dZ: torch.Tensor, add1: Optional[torch.Tensor] = None, relu_in: bool = False) -> torch.Tensor:
        """dZ = gradient w.r.t. the convolution output (before ReLU / residual); returns the gradient w.r.t. the GroupNorm input
        (+ add1).  relu_in: the layer's input r.x is a post-ReLU activation and the caller wants the gradient in front of that
        ReLU (masked by r.x > 0, with its max |.| recorded for the next dynamic scale).  Accumulates the conv weight and GroupNorm
        affine gradients."""
        m = self.mats[r.name]
        key = self.prefix + r.name
        B, D0, D1, D2, cin = r.x.shape
        cout, nvox, G, st = m["cout"], D0 * D1 * D2, m["groups"], _lib.stream()
        sc, sh, s2 = self._scale(dZ, B, cout)                # dynamic power-of-two scale of dZ, shared by the weight and data gradients
        inv = s2[1:]
        dW = self.g[key + "conv.weight"]                     # [cout, cin, 3, 3, 3]: the kernels accumulate in this layout directly
        # routing by the entry point's OWN predicate (semabs_wgrad_conv3_supported: shape, 32-bit staging offsets, scratch size), so that no shape can
        # fall into its SEMABS_REQUIRE (ADVICE rounds 3, 4); anything it does not take goes to the row kernels below
        scr = self._wg_scratch() if self.wgrad_tr else (None, 0)
        red = self.arena.zeros((B, cin, 2), torch.float64)
        have_red = False
        if self.mfma_wgrad and self.wgrad_tr and self.wgrad_gn and self._wgrad_conv3_gn_ok(B, D0, D1, D2, cout, cin, scr[1]):
            # one pass over (dZ, x): the weight gradient AND the (sum dXn, sum dXn xhat) the GroupNorm backward needs (csrc/train.hip has the algebra)
            _lib.call("semabs_wgrad_conv3_gn", _lib.ptr(dZ), _lib.ptr(r.x), _lib.ptr(r.mean), _lib.ptr(r.rstd), G, _lib.ptr(self.p[key + "groupnorm.weight"]),
                      _lib.ptr(self.p[key + "groupnorm.bias"]), _lib.ptr(self.p[key + "conv.weight"]), _lib.ptr(s2), _lib.ptr(dW), _lib.ptr(red),
                      B, D0, D1, D2, cout, cin, *scr, st)
            have_red = True
        elif self.mfma_wgrad and self._wgrad_conv3_route(D0, D1, D2, cout, cin, scr[1]):
            _lib.call("semabs_wgrad_conv3", _lib.ptr(dZ), _lib.ptr(r.x), _lib.ptr(r.scale), _lib.ptr(r.shift), _lib.ptr(s2), _lib.ptr(dW),
                      B, D0, D1, D2, cout, cin, 1, *scr, st)
        elif cin % 16 == 0 and self.mfma_wgrad:                  # 8^3 / 4^3 levels: rows through LDS, transposing reads (k_wgrad_mfma)
            _lib.call("semabs_wgrad_mfma", _lib.ptr(dZ), _lib.ptr(r.x), _lib.ptr(r.scale), _lib.ptr(r.shift), _lib.ptr(s2), None, _lib.ptr(dW),
                      B, D0, D1, D2, D0, D1, D2, 1, cout, cin, 27, TAPS_CONV3, 1, *self._wg_scratch(), st)
        else:
            _lib.call("semabs_wgrad", _lib.ptr(dZ), _lib.ptr(r.x), _lib.ptr(r.scale), _lib.ptr(r.shift), _lib.ptr(dW), B, D0, D1, D2, D0, D1, D2, 1,
                      cout, cin, 27, TAPS_CONV3, 1, st)
        if have_red and self.fuse_gn_apply and self._conv3d_gnbwd_ok(B, D0, D1, D2, cout, cin, G, add1 is not None):
            # the sums are known BEFORE the data gradient (they came out of the weight-gradient pass), so the GroupNorm backward can be the data-gradient
            # convolution's epilogue: no dXn tensor, no apply pass (read dXn, read x, write dX)
            coef = torch.empty(B, cin, 3, dtype=torch.float32, device=self.dev)
            _lib.call("semabs_gn_bwd_coef", _lib.ptr(red), _lib.ptr(self.p[key + "groupnorm.weight"]), _lib.ptr(r.rstd), _lib.ptr(inv), _lib.ptr(coef),
                      _lib.ptr(self.g[key + "groupnorm.weight"]), _lib.ptr(self.g[key + "groupnorm.bias"]), B, cin, G, nvox, st)
            dX = torch.empty(B, D0, D1, D2, cin, dtype=torch.float32, device=self.dev)
            bits = self.arena.zeros((1,), torch.int32)
            _lib.call("semabs_conv3d_gnbwd", _lib.ptr(dZ), _lib.ptr(m["bwd"][0]), _lib.ptr(m["bwd"][1]), _lib.ptr(dX), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(r.x),
                      _lib.ptr(r.mean), _lib.ptr(r.rstd), _lib.ptr(coef), G, _lib.ptr(add1), 1 if relu_in else 0, _lib.ptr(bits), B, D0, D1, D2, cout, cin,
                      1 | m["bwd"][2], st)
            dX._semabs_absmax = bits
            return dX
        dXn = torch.empty(B, D0, D1, D2, cin, dtype=torch.float32, device=self.dev)      # = s * (d loss / d GN output)
        _lib.call("semabs_conv3d", _lib.ptr(dZ), _lib.ptr(m["bwd"][0]), _lib.ptr(m["bwd"][1]), _lib.ptr(dXn), _lib.ptr(sc), _lib.ptr(sh), None, None,
                  B, D0, D1, D2, cout, cin, 3, 0, 1 | m["bwd"][2], st)
        if not have_red:
            _lib.call("semabs_chan_reduce", _lib.ptr(dXn), _lib.ptr(r.x), _lib.ptr(r.mean), _lib.ptr(r.rstd), _lib.ptr(red), B, nvox, cin, G, st)
        coef = torch.empty(B, cin, 3, dtype=torch.float32, device=self.dev)
        _lib.call("semabs_gn_bwd_coef", _lib.ptr(red), _lib.ptr(self.p[key + "groupnorm.weight"]), _lib.ptr(r.rstd), _lib.ptr(inv), _lib.ptr(coef),
                  _lib.ptr(self.g[key + "groupnorm.weight"]), _lib.ptr(self.g[key + "groupnorm.bias"]), B, cin, G, nvox, st)
        dX = torch.empty_like(dXn)
        # max |dX| always comes out of this pass (one fmax per element, one atomic per block): whoever scales dX next - the following convolution's
        # backward, or the transposed convolution's when dX leaves the block - does not read the tensor again for it (0.2 ms at 8 x 128^3 x 16)
        bits = self.arena.zeros((1,), torch.int32)
        _lib.call("semabs_gn_bwd_apply", _lib.ptr(dXn), _lib.ptr(r.x), _lib.ptr(r.mean), _lib.ptr(r.rstd), _lib.ptr(coef), _lib.ptr(add1), None,
                  _lib.ptr(r.x if relu_in else None), _lib.ptr(bits), _lib.ptr(dX), B, nvox, cin, G, st)
        dX._semabs_absmax = bits
        return dX

    def _block_bwd(self, recs, dOut, in_scale=None):
        r1, r2, r3 = recs
        if getattr(dOut, "_semabs_relu_masked", False) and in_scale is None:
            dS = dOut                                        # (the pooling backward already went through this block's final ReLU and recorded max |dS|)
        else:
            dS = self._ew(dOut, r3.y, 0, want_max=True, in_scale=in_scale)     # through the final ReLU of relu(conv3 + out1)
        dz2 = self._conv_bwd(r3, dS, relu_in=True)          # d out2, already through conv2's ReLU (r3.x = out2)
        dz1 = self._conv_bwd(r2, dz2, add1=dS, relu_in=True)   # d out1 = via conv2 + the residual branch, through conv1's ReLU
        return self._conv_bwd(r1, dz1)

    def _up_bwd(self, pre: str, xin: torch.Tensor, g: torch.Tensor):
        """ConvTranspose3d k3 s2 p1 op1 backward: g = gradient w.r.t. its output [B, 2D, 2D, 2D, cout] -> gradient w.r.t. xin."""
        m = self.mats[pre]
        key = self.prefix + pre
        st = _lib.stream()
        B, D0, D1, D2, cin = xin.shape
        cout = m["cout"]
        self._colsum(g.view(-1, cout), self.g[key + "bias"])
        sc, sh, s2 = self._scale(g, B, cout)
        if cout % 16 == 0 and self.mfma_wgrad:
            _lib.call("semabs_wgrad_mfma", _lib.ptr(xin), _lib.ptr(g), None, None, None, _lib.ptr(s2), _lib.ptr(self.g[key + "weight"]),
                      B, D0, D1, D2, 2 * D0, 2 * D1, 2 * D2, 2, cin, cout, 27, TAPS_CONV3, 1, *self._wg_scratch(), st)      # [cin, cout, 3, 3, 3] directly
        else:
            _lib.call("semabs_wgrad", _lib.ptr(xin), _lib.ptr(g), None, None, _lib.ptr(self.g[key + "weight"]), B, D0, D1, D2, 2 * D0, 2 * D1, 2 * D2, 2,
                      cin, cout, 27, TAPS_CONV3, 1, st)
        dx = torch.empty(B, D0, D1, D2, cin, dtype=torch.float32, device=self.dev)
        _lib.call("semabs_conv3d_gather", _lib.ptr(g), _lib.ptr(m["bwd"][0]), _lib.ptr(m["bwd"][1]), _lib.ptr(dx), _lib.ptr(sc), _lib.ptr(sh),
                  B, 2 * D0, 2 * D1, 2 * D2, D0, D1, D2, 2, cout, cin, 27, TAPS_CONV3, 1 | m["bwd"][2], st)
        return dx, s2[1:]                                    # still scaled: the block that consumes it multiplies by 1 / s in its first pass

    def backward(self, tape, dy: torch.Tensor, on_done=None) -> torch.Tensor:
        """dy fp32 [B, S, S, S, Cout] -> gradient w.r.t. the UNet input; parameter gradients are accumulated into `grads`.
        on_done(name): called when every parameter gradient of "decoders.<i>" / "encoders.<i>" has been issued (in that order: finest decoder first,
        encoder 0 last) - the data-parallel trainer starts that part's all-reduce behind it while the rest of the backward pass runs."""
        st = _lib.stream()
        L = len(self.f_maps)
        d_skip: Dict[int, torch.Tensor] = {}
        g = dy.contiguous()
        g_scale = None                                       # device scalar: g still carries a dynamic gradient scale (undone by its consumer)
        for item in reversed(tape):
            kind = item[0]
            if self.debug is not None:
                self.debug.append((kind, g))
            if kind == "final":
                x = item[1]
                m = self.mats["final_conv."]
                B, D0, D1, D2, cin = x.shape
                cout = m["cout"]
                R = B * D0 * D1 * D2
                _lib.call("semabs_wgrad", _lib.ptr(g), _lib.ptr(x), None, None, _lib.ptr(self.g[self.prefix + "final_conv.weight"]), 1, 1, 1, R, 1, 1, R, 1,
                          cout, cin, 1, TAPS_ONE, 1, st)
                dx = torch.empty(B, D0, D1, D2, cin, dtype=torch.float32, device=self.dev)
                sc, sh, s2 = self._scale(g, B, cout)
                wf = self.p[self.prefix + "final_conv.weight"].detach()
                fast16 = self.rows_linear and cin == 16 and cout == 16 and R >= (1 << 16) and wf.is_contiguous()       # k_rows16_f32: also sums g's columns (the bias gradient)
                if not fast16:
                    self._colsum(g.view(R, cout), self.g[self.prefix + "final_conv.bias"])
                if self.rows_linear and cout % 4 == 0 and cin <= 128 and wf.is_contiguous():      # dx = (s g) W: the transposed row-linear layer (see forward)
                    # ... scaled back on the way out, through the mask of the ReLU that produced x (the last decoder block's final ReLU), max |dx| recorded:
                    # that block's own mask pass over the 8 x 128^3 x 16 tensor (read, read, write: 0.58 ms) is not run
                    bits = self.arena.zeros((1,), torch.int32)
                    _lib.call("semabs_linear_rows", _lib.ptr(g), cout, _lib.ptr(wf), 1, cin, None, _lib.ptr(dx), R, cout, cin, 0, 0.0, _lib.ptr(s2), s2[1:].data_ptr(),
                              _lib.ptr(x), _lib.ptr(bits), None, _lib.ptr(self.g[self.prefix + "final_conv.bias"]) if fast16 else None, st)
                    dx._semabs_absmax = bits
                    dx._semabs_relu_masked = True
                    g, g_scale = dx, None
                    continue
                else:
                    _lib.call("semabs_conv3d", _lib.ptr(g), _lib.ptr(m["bwd"][0]), _lib.ptr(m["bwd"][1]), _lib.ptr(dx), _lib.ptr(sc), _lib.ptr(sh), None, None,
                              B, D0, D1, D2, cout, cin, 1, 0, 1 | m["bwd"][2], st)
                g, g_scale = dx, s2[1:]
            elif kind == "block":
                g = self._block_bwd(item[1:], g, in_scale=g_scale)
                g_scale = None
                if on_done is not None and item[1].name.startswith("encoders."):
                    on_done(".".join(item[1].name.split(".")[:2]))
            elif kind == "up":
                _, pre, xin, level = item
                assert g_scale is None
                d_skip[level] = g                                   # y = skip + convT(x) + bias: the skip gets the same gradient
                g, g_scale = self._up_bwd(pre, xin, g)
                if on_done is not None:
                    on_done(".".join(pre.split(".")[:2]))
            elif kind == "pool":
                _, x, level = item                                  # x = output of encoder `level`, also used as a skip
                assert g_scale is None
                B, D0, D1, D2, Cc = x.shape
                dx = torch.empty_like(x)
                # route the gradient to the arg-max AND add the skip's gradient, in one pass
                # ... AND apply the mask of the ReLU that produced x (= the output of the residual block whose backward comes next), with max |dx| for its
                # dynamic gradient scale: the block's own mask pass (read dx, read x, write) is not run
                bits = self.arena.zeros((1,), torch.int32)
                _lib.call("semabs_maxpool3d_bwd_add", _lib.ptr(x), _lib.ptr(g), _lib.ptr(d_skip.pop(level)), _lib.ptr(dx), _lib.ptr(bits), 1, B, D0, D1, D2, Cc, st)
                dx._semabs_absmax = bits
                dx._semabs_relu_masked = True
                g = dx
        assert not d_skip
        if g_scale is not None:                              # (a network whose first tape entry is not a block)
            g = self._unscale_by(g, g_scale)
        return g


class VOOLTrainer:
    """`SemAbsVOOL` (net.py:469-579, pointing_method "cosine_sim") + loss + optimiser as one training-step object.

        tr = VOOLTrainer(state_dict, voxel_shape=(128,)*3, scene_bounds=..., unet_f_maps=16, unet_num_levels=6, ...)
        stats = tr.step(batch)            # forward, BCE, backward, (all-reduce), clip, LAMB; returns {"loss", "gradnorm"}
        tr.state_dict()                   # fp32 tensors under the reference's key names

    `batch` carries the reference's VOOL batch keys (dataset.py / train_vool.py:118-178): input_xyz_pts [B, N, 3],
    input_target_saliency_pts / input_reference_saliency_pts [B, D, N(, 1)], output_xyz_pts [B, D, M, 3], output_label_pts [B, D, M],
    spatial_relation_name (D lists of B names).
    """

    def __init__(self, state_dict: Optional[Dict[str, torch.Tensor]], voxel_shape, scene_bounds, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8,
                 unet_num_levels=6, pts_feat_extractor_hidden_dim=128, pointing_dim=64, lr=1e-3, weight_decay=1e-5, grad_max_norm=2.0,
                 balance_positive_negative=False, pointing_temperature=0.07, module: Optional[torch.nn.Module] = None):
        """state_dict: the reference's `SemAbsVOOL.state_dict()` (fp32 master copies are made), or `module=` an nn.Module with the reference's
        parameter tree (`semabs_amd.net.SemAbsVOOL`): then the module's OWN nn.Parameters are the master weights - nothing is copied, the
        caller's optimiser updates them - and gradients are handed to autograd (`forward_tape` / `backward_tape`, used by `SemAbsVOOL.forward`)."""
        dev = self.dev = _lib.require_gpu()
        assert pointing_dim == 64 and unet_num_channels == 16, "the VOOL head kernels cover pointing_dim=64 over 16-channel volumes"
        self.vg = VirtualGrid(scene_bounds=np.array(scene_bounds), batch_size=1, grid_shape=tuple(voxel_shape))
        self.C, self.H, self.E = unet_num_channels, pts_feat_extractor_hidden_dim, pointing_dim
        self.grad_max_norm, self.balance, self.temperature = float(grad_max_norm), bool(balance_positive_negative), float(pointing_temperature)
        self.module = module
        if module is not None:
            assert state_dict is None
            self.extra = {k: v for k, v in module.named_buffers()}
            self.params = dict(module.named_parameters())
            names = list(self.params)
            for k, p_ in self.params.items():
                assert p_.is_cuda and p_.dtype == torch.float32 and p_.is_contiguous(), f"{k}: parameters must be fp32 on the HIP device"
        else:
            sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
            self.extra = {k: v.clone() for k, v in sd.items() if not torch.is_floating_point(v) or k.endswith("steps")}
            names = [k for k in sd if k not in self.extra]
            self.params = {k: torch.nn.Parameter(sd[k].detach().float().to(dev).contiguous(), requires_grad=False) for k in names}
        # parameters the VOOL graph never touches get no gradient, like p.grad = None in the reference (completion_net.visual_sampler.*)
        self.names = names
        self.trainable = [k for k in names if ".visual_sampler." not in k]
        self.total = sum(self.params[k].numel() for k in self.trainable)
        self.grads: Dict[str, torch.Tensor] = {}
        self.unet = UNetTrainer(self.params, self.grads, "completion_net.vol_feature_extractor.", unet_num_channels, unet_num_channels,
                                unet_f_maps, unet_num_groups, unet_num_levels)
        self._bind_grads(torch.zeros(self.total + len(RELATIONS), dtype=torch.float32, device=dev), attach=module is None)
        # Like the reference (`Lamb(net.parameters())`, utils.py:264-266) the optimiser spans EVERY parameter in state-dict order - the ones the
        # VOOL graph never reaches keep grad = None and are skipped by step() - so `optimizer.state_dict()` is interchangeable with the reference's
        self.opt = Lamb([self.params[k] for k in names], lr=lr, weight_decay=weight_decay) if module is None else None
        self.steps = 0
        # data-parallel gradient exchange: bucketed and overlapped with the backward pass (default) or ONE blocking all-reduce after it (A/B, tests)
        self.overlap_allreduce = os.environ.get("SEMABS_DP_OVERLAP", "1") == "1"
        self.rows_linear = os.environ.get("SEMABS_ROWS_LINEAR", "1") == "1"      # the MLP layers on semabs_linear_rows (A/B: 0 = k_linear / 1x1x1 convolutions)
        self._sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.last = {}

    def _bind_grads(self, flat: torch.Tensor, attach: bool):
        """One flat buffer: every gradient + one "used in this step" flag per relation embedding (they ride in the same all-reduce)."""
        self.flat_grad = flat
        self.rel_flags = flat[self.total:]
        off = 0
        starts = {}
        for k in self.trainable:
            n = self.params[k].numel()
            self.grads[k] = flat[off: off + n].view_as(self.params[k])
            if attach:
                self.params[k].grad = self.grads[k]
            starts[k] = off
            off += n
        # Data-parallel buckets, in the order the backward pass completes them (state-dict order = point MLP, encoders 0 .. L-1, decoders 0 .. L-2,
        # final conv, samplers, relation embeddings, flags; the backward pass runs it roughly backwards): [decoder 0 .. end of the buffer] is complete
        # when decoder 0 (the coarsest, processed last of the decoders) is, then the two deepest encoders - 96 % of the bytes - each on its own, and
        # the small head of the buffer [point MLP, encoders 0 .. L-3] at the very end.  Their all-reduces run behind the fine levels' backward kernels.
        up = "completion_net.vol_feature_extractor."
        first = lambda pre: min((o for k, o in starts.items() if k.startswith(up + pre)), default=None)
        L = len(self.unet.f_maps)
        cuts = {"decoders.0": first("decoders.0."), f"encoders.{L - 1}": first(f"encoders.{L - 1}."), f"encoders.{L - 2}": first(f"encoders.{L - 2}.")}
        self.bucket_of = {}
        ranges = []
        if all(v is not None for v in cuts.values()) and 0 < cuts[f"encoders.{L - 2}"] < cuts[f"encoders.{L - 1}"] < cuts["decoders.0"] < flat.numel():
            ranges = [(cuts["decoders.0"], flat.numel()), (cuts[f"encoders.{L - 1}"], cuts["decoders.0"]),
                      (cuts[f"encoders.{L - 2}"], cuts[f"encoders.{L - 1}"]), (0, cuts[f"encoders.{L - 2}"])]
            self.bucket_of = {"decoders.0": 0, f"encoders.{L - 1}": 1, f"encoders.{L - 2}": 2}
        else:
            ranges = [(0, flat.numel())]
        from .dist import BucketedAllReduce
        self.buckets = BucketedAllReduce(flat, ranges)

    # ---- helpers -----------------------------------------------------------------------------------------------------------------
    def _linear(self, x, w, b, act, grad_in=False):
        """grad_in: x is a gradient (values of 1e-7 and below): the matrix-core kernel then scales it by a power of two on the way in and the
        accumulator back on the way out - unscaled, such values are fp16 subnormals before the hi / lo split (the fp32 FMA kernel does not care)."""
        R, Ci = x.shape
        Co = w.shape[0]
        y = torch.empty(R, Co, dtype=torch.float32, device=self.dev)
        if self.rows_linear and Ci % 4 == 0 and Co <= 128 and w.is_contiguous():
            # (round 5) the matrix-core row kernel: the fp32 FMA kernel below took 0.73 ms per layer at 640 k - 1.6 M rows
            s2 = self.unet._scale(x, 1, Ci)[2] if grad_in else None
            _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(w), Ci, 1, _lib.ptr(b), _lib.ptr(y), R, Ci, Co, int(act), SLOPE,
                      _lib.ptr(s2), None if s2 is None else s2[1:].data_ptr(), None, None, None, None, _lib.stream())
            return y
        _lib.call("semabs_linear_f32", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), R, Ci, Co, act, SLOPE, _lib.stream())
        return y

    def _linear_mfma(self, x, wkey, b, act, grad_in=False, transposed=False, mask=None, colsum=None):
        """The 128 -> 128 MLP layers as 1x1x1 convolutions on the split-fp16 MFMA kernel (fp32-like accuracy); wkey names the weight parameter,
        transposed = multiply by its transpose (the data gradient).  grad_in: x is a gradient (arbitrarily small): scaled by a power of two on
        the way in and back on the way out.  mask (with grad_in): the layer input's pre-activation sign source h - the result is the UNSCALED gradient in
        front of that LeakyReLU (y * (h > 0 ? 1 : SLOPE)) with its max |.| recorded - and its column sums added to `colsum` (the bias gradient of the layer
        that produced h) - in the same pass where the row kernel runs."""
        R, Ci = x.shape
        w = self.params[wkey]
        Co = w.shape[1] if transposed else w.shape[0]
        if mask is not None:
            assert grad_in and b is None and not act
            if self.rows_linear and Ci % 4 == 0 and Co <= 128:
                y = torch.empty(R, Co, dtype=torch.float32, device=self.dev)
                s2 = self.unet._scale(x, 1, Ci)[2]
                bits = self.unet.arena.zeros((1,), torch.int32)
                wd = w.detach()
                _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(wd), 1 if transposed else wd.shape[1], wd.shape[1] if transposed else 1,
                          None, _lib.ptr(y), R, Ci, Co, 2, SLOPE, _lib.ptr(s2), s2[1:].data_ptr(), _lib.ptr(mask), _lib.ptr(bits), _lib.ptr(colsum), None, _lib.stream())
                y._semabs_absmax = bits
                return y
            y_, inv_ = self._linear_mfma(x, wkey, None, 0, grad_in=True, transposed=transposed)
            y = self.unet._ew(y_, mask, 1, want_max=True, in_scale=inv_)
            if colsum is not None:
                self.unet._colsum(y, colsum)
            return y
        if self.rows_linear and Ci % 4 == 0 and Co <= 128:
            # (round 5) the matrix-core row kernel reads W (or its transpose, through strides) straight from the fp32 parameter and splits it while
            # staging: no per-step operand gather, and 0.05 - 0.1 ms per layer where the 1 x 1 x 1 "convolution" on the gather kernel took 0.7 - 0.9 ms
            y = torch.empty(R, Co, dtype=torch.float32, device=self.dev)
            s2 = self.unet._scale(x, 1, Ci)[2] if grad_in else None
            wd = w.detach()
            _lib.call("semabs_linear_rows", _lib.ptr(x), Ci, _lib.ptr(wd), 1 if transposed else wd.shape[1], wd.shape[1] if transposed else 1,
                      _lib.ptr(b), _lib.ptr(y), R, Ci, Co, 1 if act else 0, SLOPE, _lib.ptr(s2), None, None, None, None, None, _lib.stream())
            return (y, s2[1:]) if grad_in else y
        hi, lo, pk = self.unet.layouts.get(f"lin:{wkey}:{int(transposed)}", w, (lambda t: _flat_packed(_pad32(t.t().contiguous()))) if transposed else
                                           (lambda t: _flat_packed(_pad32(t.contiguous()))))
        y = torch.empty(R, Co, dtype=torch.float32, device=self.dev)
        u = self.unet
        sc = sh = s2 = None
        if grad_in:
            sc, sh, s2 = u._scale(x, 1, Ci)
        _lib.call("semabs_conv3d", _lib.ptr(x), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(y), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(b), None,
                  1, 1, 1, R, Ci, Co, 1, 2 if act else 0, 1 | pk, _lib.stream())
        return (y, s2[1:]) if grad_in else y                   # grad_in: still scaled; the caller's LeakyReLU-mask pass multiplies by 1 / s

    def _wgrad_linear(self, dOut, x, grad_w, cols=None):
        R, Co = dOut.shape
        Ci = x.shape[1]
        if cols is None:
            dW = grad_w                                           # [Co, Ci]: accumulate in place
        else:
            dW = self.unet.arena.zeros((Co, Ci), torch.float32)
        if Ci % 16 == 0 and self.unet.mfma_wgrad:
            s2 = self.unet._scale(dOut, 1, Co)[2]
            _lib.call("semabs_wgrad_mfma", _lib.ptr(dOut), _lib.ptr(x), None, None, _lib.ptr(s2), None, _lib.ptr(dW), 1, 1, 1, R, 1, 1, R, 1, Co, Ci, 1,
                      TAPS_ONE, 1, *self.unet._wg_scratch(), _lib.stream())
        else:
            _lib.call("semabs_wgrad", _lib.ptr(dOut), _lib.ptr(x), None, None, _lib.ptr(dW), 1, 1, 1, R, 1, 1, R, 1, Co, Ci, 1, TAPS_ONE, 1, _lib.stream())
        if cols is not None:
            grad_w.add_(dW[:, :cols])

    def bce_weight(self, label: torch.Tensor) -> Optional[torch.Tensor]:
        """utils.get_bce_weight (utils.py:727-749); label [B, D, M] fp32 on the device.  None = all ones."""
        if not self.balance:
            return None
        w = torch.ones_like(label)
        pos = label.bool()
        pp = pos.float().mean(dim=2, keepdim=True)
        w = torch.where(pos, 1.0 / (pp + 1e-10), 1.0 / ((1 - pp) + 1e-10))
        return (w * (label.numel() / w.sum())).contiguous()

    # ---- forward / backward of one scene -------------------------------------------------------------------------------------------
    def _scene_fwd(self, xyz, sal_t, sal_r, query, rel_names) -> dict:
        """Forward up to the pointer head's input `o` [D*M, 64]; returns the tape the backward walks."""
        dev, st, u = self.dev, _lib.stream(), self.unet
        p = self.params
        D, N = sal_t.shape
        M = query.shape[1]
        P = 2 * D
        S0, S1, S2 = self.vg.grid_shape
        nvox = S0 * S1 * S2
        cn = "completion_net.pts_feat_extractor."
        # point MLP (net.py:395-404) for the 2 D saliency channels at once: target volumes first, reference volumes second
        feat = torch.cat([sal_t, sal_r], dim=0)                                           # [P, N]
        x4 = torch.cat([xyz.unsqueeze(0).expand(P, N, 3), feat.unsqueeze(-1)], dim=-1).reshape(P * N, 4).contiguous()
        h1 = self._linear(x4, p[cn + "0.weight"], p[cn + "0.bias"], 1)
        h2 = self._linear_mfma(h1, cn + "2.weight", p[cn + "2.bias"], 1)
        pf = self._linear_mfma(h2, cn + "4.weight", p[cn + "4.bias"], 0)              # [P*N, C]
        flat = self.vg.flat_idxs(xyz)
        vol = torch.zeros(P, S0, S1, S2, self.C, dtype=torch.float32, device=dev)
        head = torch.full((nvox,), -1, dtype=torch.int32, device=dev)
        nxt = torch.empty(N, dtype=torch.int32, device=dev)
        sums = None
        if self.C == 16 and u.mats["encoders.0.basic_module.conv1."]["groups"] == 8:      # first GroupNorm's statistics come out of the scatter
            sums = u.arena.zeros((P, 8, 2), torch.float64)
            _lib.call("semabs_scatter_mean_stats", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox, 1,
                      _lib.ptr(sums), st)
        else:
            _lib.call("semabs_scatter_mean", _lib.ptr(flat), _lib.ptr(pf), _lib.ptr(head), _lib.ptr(nxt), _lib.ptr(vol), P, N, self.C, nvox, 1, st)
        fv, tape = u.forward(vol, in_sums=sums)
        # VOOL head (net.py:556-579)
        off3, sc3, shp = _lib.farr(self.vg.offsets), _lib.farr(self.vg.scales), _lib.iarr(self.vg.grid_shape)
        f = torch.empty(D * M, 36, dtype=torch.float32, device=dev)
        _lib.call("semabs_vool_sample", _lib.ptr(fv[:D]), _lib.ptr(fv[D:]), _lib.ptr(query), off3, sc3, shp, D, M, _lib.ptr(f), st)
        ss = "spatial_sampler.mlp."
        w1p = torch.zeros(32, 36, dtype=torch.float32, device=dev)
        w1p[:, :35] = p[ss + "0.weight"]
        h = self._linear(f, w1p, p[ss + "0.bias"], 1)
        o = self._linear_mfma(h, ss + "2.weight", p[ss + "2.bias"], 0)
        rel = torch.stack([p["relation_embeddings." + n].detach() for n in rel_names], dim=0).contiguous()
        return dict(D=D, N=N, M=M, P=P, x4=x4, h1=h1, h2=h2, flat=flat, tape=tape, query=query, f=f, w1p=w1p, h=h, o=o, rel=rel,
                    rel_names=list(rel_names))

    def _scene_bwd(self, c: dict, dO: torch.Tensor, drel: torch.Tensor):
        """dO = d loss / d o [D*M, 64], drel = d loss / d (relation embedding rows) [D, 64]: accumulates every parameter gradient."""
        dev, st, u = self.dev, _lib.stream(), self.unet
        p, g = self.params, self.grads
        D, N, M, P = c["D"], c["N"], c["M"], c["P"]
        S0, S1, S2 = self.vg.grid_shape
        nvox = S0 * S1 * S2
        cn, ss = "completion_net.pts_feat_extractor.", "spatial_sampler.mlp."
        off3, sc3, shp = _lib.farr(self.vg.offsets), _lib.farr(self.vg.scales), _lib.iarr(self.vg.grid_shape)
        h, f, w1p, query, flat, h1, h2, x4 = c["h"], c["f"], c["w1p"], c["query"], c["flat"], c["h1"], c["h2"], c["x4"]
        for d, n in enumerate(c["rel_names"]):
            g["relation_embeddings." + n].add_(drel[d])
        self._wgrad_linear(dO, h, g[ss + "2.weight"])
        dob = getattr(dO, "_semabs_colsum", None)
        if dob is not None:
            g[ss + "2.bias"].add_(dob.sum(0))                # (came out of the loss kernel that wrote dO)
        else:
            u._colsum(dO, g[ss + "2.bias"])
        dh = self._linear_mfma(dO, ss + "2.weight", None, 0, grad_in=True, transposed=True, mask=h, colsum=g[ss + "0.bias"])
        self._wgrad_linear(dh, f, g[ss + "0.weight"], cols=35)
        df = self._linear(dh, w1p.t().contiguous(), None, 0, grad_in=True)               # [D*M, 36]
        dvol = torch.empty(P, S0, S1, S2, self.C, dtype=torch.float32, device=dev)
        cell_head = torch.empty(D * nvox, dtype=torch.int32, device=dev)
        cell_next = torch.empty(D * M, dtype=torch.int32, device=dev)
        bits = u.arena.zeros((1,), torch.int32)
        _lib.call("semabs_vool_sample_bwd", _lib.ptr(df), _lib.ptr(query), off3, sc3, shp, D, M, _lib.ptr(cell_head), _lib.ptr(cell_next),
                  _lib.ptr(dvol[:D]), _lib.ptr(dvol[D:]), _lib.ptr(bits), st)
        dvol._semabs_absmax = bits                           # (the final convolution's backward scales dvol: no pass over it for its maximum)
        dscat = u.backward(c["tape"], dvol, on_done=c.get("on_done"))
        count = torch.zeros(nvox, dtype=torch.int32, device=dev)
        dpf = torch.empty(P * N, self.C, dtype=torch.float32, device=dev)
        _lib.call("semabs_scatter_mean_bwd", _lib.ptr(flat), _lib.ptr(count), _lib.ptr(dscat), _lib.ptr(dpf), P, N, self.C, nvox, st)
        self._wgrad_linear(dpf, h2, g[cn + "4.weight"])
        u._colsum(dpf, g[cn + "4.bias"])
        dh2 = self._linear_mfma(dpf, cn + "4.weight", None, 0, grad_in=True, transposed=True, mask=h2, colsum=g[cn + "2.bias"])
        self._wgrad_linear(dh2, h1, g[cn + "2.weight"])
        dh1 = self._linear_mfma(dh2, cn + "2.weight", None, 0, grad_in=True, transposed=True, mask=h1, colsum=g[cn + "0.bias"])
        self._wgrad_linear(dh1, x4, g[cn + "0.weight"])

    def _scene(self, xyz, sal_t, sal_r, query, label, weight, rel_names, n_total, loss_acc, logits_out, last=False):
        """Fused form (VOOLTrainer.step): the BCE-with-logits loss and its gradient come out of the pointer-head kernel.
        last: this is the step's last scene - its backward pass announces the finished gradient buckets to the data-parallel all-reduce."""
        c = self._scene_fwd(xyz, sal_t, sal_r, query, rel_names)
        if last and self.overlap_allreduce and self.buckets.active:
            c["on_done"] = lambda name: self.buckets.ready(self.bucket_of[name]) if name in self.bucket_of else None
        dO = torch.empty_like(c["o"])
        drel = self.unet.arena.zeros((c["D"], self.E), torch.float32)
        dob = self.unet.arena.zeros((c["D"], self.E), torch.float32)           # column sums of dO per description: the sampler MLP's last bias gradient
        _lib.call("semabs_cos_bce", _lib.ptr(c["o"]), _lib.ptr(c["rel"]), _lib.ptr(label), _lib.ptr(weight), c["D"], c["M"], self.temperature, n_total,
                  _lib.ptr(logits_out), _lib.ptr(dO), _lib.ptr(drel), _lib.ptr(loss_acc), _lib.ptr(dob), _lib.stream())
        dO._semabs_colsum = dob
        self._scene_bwd(c, dO, drel)

    # ---- autograd boundary (SemAbsVOOL.forward under grad mode) ---------------------------------------------------------------------
    def _unpack(self, batch: dict):
        dev = self.dev
        xyz = batch["input_xyz_pts"].to(dev, torch.float32)
        B, N = xyz.shape[:2]
        st_ = batch["input_target_saliency_pts"].to(dev, torch.float32).reshape(B, -1, N)
        sr_ = batch["input_reference_saliency_pts"].to(dev, torch.float32).reshape(B, -1, N)
        D = st_.shape[1]
        q = batch["output_xyz_pts"].to(dev, torch.float32).reshape(B, D, -1, 3)
        names = np.array(batch["spatial_relation_name"]).T.reshape(B, D)                 # [B, D] like net.py:527
        return xyz, st_, sr_, q, names, B, N, D, int(q.shape[2])

    @torch.no_grad()
    def forward_tape(self, batch: dict):
        """-> (logits [B, D, M], ctx).  No loss: the caller computes it from the logits (the reference's `get_losses`, train_vool.py:118-178)."""
        self.unet.refresh()
        self.unet.arena.reset()
        # One outstanding tape per engine: refresh() rewrites the persistent weight layouts in place and reset() recycles the arena
        # the saved activations live in, so a tape is only valid until the next forward and only while the parameters it was taken with are
        # unchanged.  The tape carries a generation number and the parameters' version counters; backward_tape refuses a stale one.
        self._tape_gen = getattr(self, "_tape_gen", 0) + 1
        xyz, st_, sr_, q, names, B, N, D, M = self._unpack(batch)
        logits = torch.empty(B, D, M, dtype=torch.float32, device=self.dev)
        scenes = []
        for b in range(B):
            c = self._scene_fwd(xyz[b].contiguous(), st_[b].contiguous(), sr_[b].contiguous(), q[b].contiguous(), list(names[b]))
            _lib.call("semabs_cos_head", _lib.ptr(c["o"]), _lib.ptr(c["rel"]), None, D, M, self.temperature, _lib.ptr(logits[b]), None, None, _lib.stream())
            scenes.append(c)
        return logits, dict(scenes=scenes, used=sorted(set(names.reshape(-1).tolist()), key=RELATIONS.index), gen=self._tape_gen, versions=self._param_versions())

    def _param_versions(self):
        return tuple(int(t._version) for t in self.params.values())

    @torch.no_grad()
    def backward_tape(self, ctx: dict, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        """dlogits = d loss / d logits [B, D, M] -> {parameter name: gradient} for every parameter the graph reached (views of ONE fresh flat
        buffer; autograd accumulates them into `.grad`, where DistributedDataParallel's hooks pick them up)."""
        if ctx.get("gen") != getattr(self, "_tape_gen", None) or not ctx.get("scenes") or any(len(c) == 0 for c in ctx["scenes"]):
            raise RuntimeError("SemAbsVOOL backward: this forward's tape is stale - a later forward of the same module has recycled its buffers, or it "
                               "was already back-propagated (one outstanding forward per engine, single-use tape; see INTEGRATION.md)")
        if ctx.get("versions") != self._param_versions():
            raise RuntimeError("SemAbsVOOL backward: parameters were modified (optimizer.step / in-place edit) between this forward and its backward; "
                               "the data gradients would be computed with the updated weights")
        self._bind_grads(torch.zeros(self.total + len(RELATIONS), dtype=torch.float32, device=self.dev), attach=False)
        dl = dlogits.to(self.dev, torch.float32).contiguous()
        for b, c in enumerate(ctx["scenes"]):
            dO = torch.empty_like(c["o"])
            drel = self.unet.arena.zeros((c["D"], self.E), torch.float32)
            _lib.call("semabs_cos_head", _lib.ptr(c["o"]), _lib.ptr(c["rel"]), _lib.ptr(dl[b]), c["D"], c["M"], self.temperature, None, _lib.ptr(dO),
                      _lib.ptr(drel), _lib.stream())
            self._scene_bwd(c, dO, drel)
            c.clear()                                                        # the tape is single-use (retain_graph is not supported)
        return {k: self.grads[k] for k in self.graph_params(ctx["used"])}

    def graph_params(self, used_relations) -> List[str]:
        """Names of the parameters an autograd graph of this forward contains: everything but `visual_sampler.*` and the relation embeddings
        no description of the batch names (they end the step with grad = None in the reference too)."""
        used = set(used_relations)
        return [k for k in self.trainable if not k.startswith("relation_embeddings.") or k[len("relation_embeddings."):] in used]

    # ---- public ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_backward(self, batch: dict) -> dict:
        """Zero the gradients, run forward + loss + backward over the batch; returns {"loss": device scalar, "logits": [B, D, M]}."""
        dev = self.dev
        assert self.module is None, "module-bound engines hand their gradients to autograd (SemAbsVOOL.forward under grad mode)"
        self.flat_grad.zero_()
        self.unet.refresh()
        self.unet.arena.reset()
        xyz, st_, sr_, q, names, B, N, D, M = self._unpack(batch)
        label = batch["output_label_pts"].to(dev, torch.float32).contiguous()
        weight = self.bce_weight(label)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)             # (its own tensor: the caller keeps it across steps, the arena is recycled)
        logits = torch.empty(B, D, M, dtype=torch.float32, device=dev)
        # like autograd, a relation embedding no description of the batch uses ends the step with grad = None, so Lamb skips it
        # entirely (no weight decay either); under DDP "used" means used on any rank (find_unused_parameters=True, utils.py:257)
        used = set(names.reshape(-1).tolist())
        self.rel_flags.copy_(torch.tensor([1.0 if n in used else 0.0 for n in RELATIONS]))
        for n in RELATIONS:                                                  # local view; optimizer_step() widens it to "any rank"
            if "relation_embeddings." + n in self.grads:
                self.params["relation_embeddings." + n].grad = self.grads["relation_embeddings." + n] if n in used else None
        self.buckets.begin_step()
        for b in range(B):
            self._scene(xyz[b].contiguous(), st_[b].contiguous(), sr_[b].contiguous(), q[b].reshape(D, M, 3).contiguous(), label[b],
                        None if weight is None else weight[b], list(names[b]), B * D * M, loss, logits[b], last=(b == B - 1))
        return {"loss": loss[0], "logits": logits}

    @torch.no_grad()
    def optimizer_step(self) -> torch.Tensor:
        """(all-reduce ->) clip_grad_norm_ -> Lamb.step; returns the pre-clip global gradient norm (device scalar)."""
        if self.overlap_allreduce:
            # the buckets the backward pass announced are already in flight (RCCL); the rest start now; the compute stream then waits for all of them
            scale = self.buckets.finish()
            used = self.flat_grad[self.flat_grad.numel() - len(RELATIONS):].detach().cpu() > 0
        else:
            from .dist import allreduce_flat_gradients
            scale, used = allreduce_flat_gradients(self.flat_grad, len(RELATIONS))
        for n, u_ in zip(RELATIONS, used.tolist()):
            k = "relation_embeddings." + n
            if k in self.grads:
                self.params[k].grad = self.grads[k] if u_ else None
        opt = self.opt
        ps = [p for p in opt.param_groups[0]["params"] if p.grad is not None]
        plan = opt._build_plan(ps, self.dev)
        _lib.call("semabs_clip_grad_norm", _lib.ptr(plan["chunks"]), plan["n_chunks"], _lib.ptr(plan["ptrs"]), len(ps), self.grad_max_norm,
                  scale, _lib.ptr(self._sq), _lib.stream())
        opt.step()
        self.steps += 1
        return torch.sqrt(self._sq[0]) * scale

    def step(self, batch: dict) -> dict:
        out = self.forward_backward(batch)
        out["gradnorm"] = self.optimizer_step()
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {k: v.detach().clone() for k, v in self.params.items()}
        sd.update({k: v.clone() for k, v in self.extra.items()})
        if "steps" in sd:                                    # only the top-level counter advances (utils.py:417 `net.steps += 1`); completion_net.steps stays
            sd["steps"] = sd["steps"] + self.steps
        return sd

    def checkpoint(self, epoch: int = 0) -> dict:
        """The reference's checkpoint dict (utils.py:278-296): {"net": state_dict, "optimizer": Lamb.state_dict(), "epochs": int}."""
        return {"net": self.state_dict(), "optimizer": self.opt.state_dict(), "epochs": int(epoch)}

    def load_checkpoint(self, ckpt: dict) -> int:
        """Restore parameters (in place: the kernels' launch plans keep pointing at the same tensors), the step counter and the optimizer
        moments; returns the stored epoch.  Accepts DDP-saved nets ("module." prefix, utils.py:283-287)."""
        net = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt["net"].items()}
        with torch.no_grad():
            for k, p in self.params.items():
                p.copy_(net[k].to(p.device, p.dtype))
        base = float(self.extra["steps"].item()) if "steps" in self.extra else 0.0
        self.steps = int(round(float(net["steps"].item()) - base)) if "steps" in net else 0
        if ckpt.get("optimizer") is not None:
            self.opt.load_state_dict(ckpt["optimizer"])
        self.unet.refresh()
        return int(ckpt.get("epochs", 0))
