"""Host orchestration of the CLIP ViT forward + closed-form attention x gradient rollout on the HIP kernels.

Mirrors what `ClipGradcam.forward` + `interpret` compute (CLIP/clip/clip_gradcam.py:58-132) for the hooked
VisionTransformer (CLIP/clip/model_explainability.py:291-355), without autograd:

* blocks 0..L-2 run for all tokens (LN -> QKV GEMM -> fused attention -> out-proj(+residual) -> LN -> c_fc+QuickGELU
  GEMM -> c_proj(+residual));
* the last block only needs what reaches the image feature `ln_post(x[:, 0]) @ proj`: K and V for all tokens, Q /
  attention / out-proj / MLP for the CLS token only (the other rows never influence the output), its softmax row is
  kept in fp32;
* for ViT-B the rollout keeps only that block (`i <= num_layers(10)` are skipped, clip_gradcam.py:85-87), so
  rel[l, n, j-1] = mean_h clamp(A[n,h,0,j] * (V[n,h,j,:] . (W_o^T g1[l,n])[h]), 0) with g1 the VJP of logit_l through
  L2-normalise -> proj -> ln_post -> CLS-token MLP + residual: six batched GEMMs with [L*n] rows.

All arithmetic is in libsemabs_hip.so; this file only owns buffers, weights and launch order.
"""
from __future__ import annotations

import math

import numpy as np
import os

import torch

from .. import _lib

EPI_F16, EPI_GELU_F16, EPI_RESID_F32, EPI_F32, EPI_ROWMAP, EPI_GELUBWD = 0, 1, 2, 3, 4, 5


class GemmTimer:
    """Optional live timing of every GEMM launch with HIP events on the launch stream (bench.py's roofline leg).  The events are filled
    by the dispatch packet of the kernel itself (`semabs_gemm_f16_ex(..., start_event, stop_event)` -> hipExtLaunchKernelGGL), not recorded around it: no barrier
    packets are inserted between consecutive kernels, so timing does not perturb the step being timed."""

    def __init__(self, every: int = 1):
        self.records = []          # (start_event, stop_event, flops, algorithmic bytes, (N, K, epilogue)); raw hipEvent_t handles
        self._free = []
        self.every = max(1, int(every))    # time one launch in `every` (hashed launch index: a uniform sample of the launch sequence)
        self.seen = 0

    def _pair(self):
        import ctypes as C
        if self._free:
            return self._free.pop()
        a, b = C.c_void_p(), C.c_void_p()
        _lib.call("semabs_event_create", C.byref(a))
        _lib.call("semabs_event_create", C.byref(b))
        return a, b

    def summary(self):
        import ctypes as C
        torch.cuda.synchronize()
        ms, tmp = 0.0, C.c_float()
        shapes = {}                # (N, K, epilogue) -> [launches, ms, flops, algorithmic bytes]
        for a, b, f, by, key in self.records:
            _lib.call("semabs_event_elapsed_ms", a, b, C.byref(tmp))
            ms += tmp.value
            d = shapes.setdefault(key, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += tmp.value; d[2] += f; d[3] += by
        fl = sum(r[2] for r in self.records)
        out = dict(launches=len(self.records), total_ms=ms, flops=fl, seen=self.seen, bytes=sum(r[3] for r in self.records), shapes=shapes)
        self._free.extend((r[0], r[1]) for r in self.records)
        self.records = []
        return out

    def __del__(self):
        try:
            for a, b in self._free + [(r[0], r[1]) for r in self.records]:
                _lib.call("semabs_event_destroy", a)
                _lib.call("semabs_event_destroy", b)
        except Exception:
            pass


GEMM_TIMER: "GemmTimer | None" = None
# Alternative residual update (off): the out-proj / c_proj GEMMs write fp16 deltas and the next LayerNorm pass adds them to the fp32 stream
# (semabs_add_layernorm).  Measured on MI355X: the GEMMs get faster (870 -> 950 TFLOP/s average, the fp32 read-modify-write leaves their
# epilogue) but the LayerNorm passes get slower by the same bytes - 149.9 vs 149.0 ms per scene - and every residual update is rounded to
# fp16 once more, so the fused read-modify-write epilogue stays the default.
DELTA_RESIDUAL = os.environ.get("SEMABS_DELTA_RESIDUAL", "0") == "1"
FUSE_GELU_BWD = os.environ.get("SEMABS_FUSE_GELU_BWD", "1") == "1"      # A/B: 0 = fp32 GEMM output + k_gelu_bwd


def gemm(A, B, C, bias, M, N, K, lda, ldb, ldc, epi, addend=None, rowmap=None, kernel=0, split=False):
    """kernel: 0 = the library's heuristic, 1 = ring kernel, 2 = phased 256 x 256 kernel (per call; tests pin a path with it).
    split: A is an [hi | lo] operand pair (K = 2 x the layer's width): the timer books the ALGORITHMIC flops (half of the issued ones)."""
    t = GEMM_TIMER
    e0 = e1 = None
    if t is not None:
        t.seen += 1
        if t.every == 1 or ((t.seen * 2654435761) >> 7) % t.every == 0:     # hashed: no phase lock with the launch pattern
            e0, e1 = t._pair()
            t.records.append((e0, e1, (1.0 if split else 2.0) * M * N * K, M * K * 2 + N * K * 2 + M * N * (2 if epi in (0, 1, 5) else (8 if epi == 2 else 4)), (int(N), int(K), int(epi))))   # A + W + C (fp32 residual: read + write)
    _lib.call("semabs_gemm_f16_ex", _lib.ptr(A), _lib.ptr(B), _lib.ptr(C), _lib.ptr(bias), _lib.ptr(addend), int(M), int(N),
              int(K), int(lda), int(ldb), int(ldc), int(epi), _lib.iarr(rowmap) if rowmap is not None else None, int(kernel), e0, e1,
              _lib.stream())


_gemm = gemm


def gemm_ln(A, B, C, bias, M, N, K, lda, ldb, ldc, epi, xg=None, gamma=None, part=None, rowac=None, colsum=None, reverse=0, lo=None, lo_cols=0, center=None):
    """A GEMM with the LayerNorm that precedes (consumer: rowac, colsum) or follows (producer: xg, gamma, part) it folded in - see
    `semabs_gemm_f16_ln` (csrc/gemm.hip, LNP / LNC).  Timed by GEMM_TIMER like every other GEMM launch; the producer's algorithmic bytes include
    the fp16 copy it writes."""
    t = GEMM_TIMER
    e0 = e1 = None
    if t is not None:
        t.seen += 1
        if t.every == 1 or ((t.seen * 2654435761) >> 7) % t.every == 0:
            e0, e1 = t._pair()
            t.records.append((e0, e1, 2.0 * M * N * K, M * K * 2 + N * K * 2 + M * N * (2 if epi in (0, 1) else 8) + (M * N * 2 if xg is not None else 0)
                              + (M * lo_cols * 2 if lo is not None else 0), (int(N), int(K), int(epi))))
    _lib.call("semabs_gemm_f16_ln", _lib.ptr(A), _lib.ptr(B), _lib.ptr(C), _lib.ptr(bias), int(M), int(N), int(K), int(lda), int(ldb), int(ldc), int(epi),
              _lib.ptr(xg), _lib.ptr(gamma), _lib.ptr(part), _lib.ptr(center), _lib.ptr(rowac), _lib.ptr(colsum), _lib.ptr(lo), int(lo_cols), int(lo.shape[1]) if lo is not None else 0,
              int(reverse), e0, e1, _lib.stream())


def ln_rowstats(part, M, ntile, D, rowac, eps=1e-5, center=None, center_out=None):
    """center: the per-row centres the producer subtracted (None = 0); center_out (may be `center`): the rows' true means = the next producer's centres."""
    _lib.call("semabs_ln_rowstats", _lib.ptr(part), int(M), int(ntile), int(D), float(eps), _lib.ptr(rowac), _lib.ptr(center), _lib.ptr(center_out), _lib.stream())


def layernorm(x, gamma, beta, out, M, D, out_f32=False, ld_in=None, eps=1e-5, order=0, mean_out=None):
    """order: 0 = rows in dispatch order, 1 / 2 = XCD-contiguous runs of rows walked forwards / backwards (the zigzag schedule of the trunk).
    mean_out fp32 [M] (optional): the row means."""
    _lib.call("semabs_layernorm", _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(out), int(M), int(D), float(eps),
              int(out_f32) | (int(order) << 1), int(ld_in if ld_in is not None else D), _lib.ptr(mean_out), _lib.stream())


def add_layernorm(x, delta, gamma, beta, out, M, D, eps=1e-5):
    _lib.call("semabs_add_layernorm", _lib.ptr(x), _lib.ptr(delta), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(out), int(M), int(D), float(eps),
              _lib.stream())


def _interp_pos_emb(pe: torch.Tensor, T: int) -> torch.Tensor:
    """The reference's hard-coded-50 positional "interpolation" (CLIP/clip/auxiliary.py:24-38), taken whenever
    T != 50 (model_explainability.py:339-343).  A [T, D] table built once at weight-load time on the host."""
    out = torch.zeros(T, pe.shape[1], dtype=pe.dtype)
    for i in range(T):
        i3 = float(i) / (T / 50)
        i1, i2 = math.floor(i3), math.ceil(i3)
        out[i] = torch.lerp(pe[i1], pe[i2], i3 - i1) if i2 < len(pe) else pe[-1]
    return out


class _BlockWeights:
    def __init__(self, sd, pre, D, heads, dev, last=False, bwd=False):
        dh = D // heads
        w_in = sd[pre + "attn.in_proj_weight"].float().clone()
        b_in = sd[pre + "attn.in_proj_bias"].float().clone()
        # q = (x W_q^T + b_q) * dh^-0.5 (auxiliary.py:207): dh^-0.5 = 2^-3 for dh = 64, folding it into W_q / b_q is exact
        scaling = float(dh) ** -0.5
        w_in[:D] *= scaling
        b_in[:D] *= scaling
        h = lambda t: t.to(dev, torch.float16).contiguous()
        f = lambda t: t.to(dev, torch.float32).contiguous()
        self.w_in, self.b_in = h(w_in), f(b_in)
        self.w_o, self.b_o = h(sd[pre + "attn.out_proj.weight"]), f(sd[pre + "attn.out_proj.bias"])
        self.ln1_w, self.ln1_b = f(sd[pre + "ln_1.weight"]), f(sd[pre + "ln_1.bias"])
        self.ln2_w, self.ln2_b = f(sd[pre + "ln_2.weight"]), f(sd[pre + "ln_2.bias"])
        self.w_fc, self.b_fc = h(sd[pre + "mlp.c_fc.weight"]), f(sd[pre + "mlp.c_fc.bias"])
        self.w_pr, self.b_pr = h(sd[pre + "mlp.c_proj.weight"]), f(sd[pre + "mlp.c_proj.bias"])
        # LayerNorm folded into the GEMM that consumes it (gemm.hip LNC): with xg = fp16(x * gamma) as the A operand,
        #   LN(x) W^T + b = rstd * (xg W^T) - mean * rstd * colsum + (b + W beta),   colsum[n] = sum_k gamma_k W[n, k]
        # (sums in fp64 over the fp16-exact weights the kernels read)
        wi, wf = self.w_in.double().cpu(), self.w_fc.double().cpu()
        g1, be1 = sd[pre + "ln_1.weight"].double().cpu(), sd[pre + "ln_1.bias"].double().cpu()
        g2, be2 = sd[pre + "ln_2.weight"].double().cpu(), sd[pre + "ln_2.bias"].double().cpu()
        self.cs_in, self.b_in_f = f(wi @ g1), f(b_in.double().cpu() + wi @ be1)
        self.cs_fc, self.b_fc_f = f(wf @ g2), f(sd[pre + "mlp.c_fc.bias"].double().cpu() + wf @ be2)
        if last or bwd:  # transposed copies: B operands ([N, K]) of the VJP GEMMs
            self.w_o_t = h(sd[pre + "attn.out_proj.weight"].float().t())
            self.w_fc_t = h(sd[pre + "mlp.c_fc.weight"].float().t())
            self.w_pr_t = h(sd[pre + "mlp.c_proj.weight"].float().t())
        if last:         # [W | W] along K: B operands of the GEMMs whose A operand is a split [hi | lo] row (vit.hip store_split4)
            two = lambda t: torch.cat([t, t], dim=1).contiguous()
            self.w_q2, self.w_k2 = two(self.w_in[:D]), two(self.w_in[D:2 * D])
            self.w_o2, self.w_fc2, self.w_pr2 = two(self.w_o), two(self.w_fc), two(self.w_pr)
            self.w_o_t2, self.w_fc_t2, self.w_pr_t2 = two(self.w_o_t), two(self.w_fc_t), two(self.w_pr_t)
        if bwd:          # d(ln_1 out) = dqkv . W_in (q rows carry the folded 1 / sqrt(dh), like the forward)
            self.w_in_t = h(w_in.t())


def make_vision_engine(state_dict, chunk_tiles: int = 2448, max_labels: int = 16):
    """ViT-B (12 blocks: only the last one enters the rollout -> closed form) or a deeper tower (ViT-L/14: 13 blocks -> `VisionRolloutDeep`)."""
    layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    if layers <= ROLLOUT_SKIP + 2:
        return VisionRollout(state_dict, chunk_tiles=chunk_tiles, max_labels=max_labels)
    # tiles per chunk of the deep rollout: the backward pass runs labels x tiles sequences (16 x 63 x 257 rows = 1 012 row panels: 15.8 waves of
    # 256 x 256 GEMM tiles at N = 1 024, where 16 tiles gave 4.02 waves, i.e. a fifth round for four tiles); 21 GB of tape at 16 labels
    # The default chunk follows the FREE device memory (a fixed 63 is a hard out-of-memory regression on smaller parts or next to a
    # resident training engine): ~0.35 GB of tape + workspaces per tile at 16 labels, keep half of what is free for everything else.
    deep = os.environ.get("SEMABS_DEEP_CHUNK")
    if deep is None:
        free, _ = torch.cuda.mem_get_info()
        per_tile = 0.35e9 * max(1, max_labels) / 16
        deep = max(4, min(63, int(0.5 * free / per_tile)))
    return VisionRolloutDeep(state_dict, chunk_tiles=min(chunk_tiles, int(deep)), max_labels=max_labels)


ROLLOUT_SKIP = 10      # ClipGradcam(num_layers=10): blocks with index <= 10 do not enter the rollout (clip_gradcam.py:37, 85-87)


class VisionRollout:
    """ViT-B image tower + last-block rollout for chunks of tiles."""

    def __init__(self, state_dict, heads: int = 12, chunk_tiles: int = 2448, max_labels: int = 16):
        dev = self.dev = _lib.require_gpu()
        sd = state_dict
        w = sd["visual.conv1.weight"]
        self.D, self.p = int(w.shape[0]), int(w.shape[-1])
        self.g = 224 // self.p
        self.T = self.g * self.g + 1
        self.H = heads
        self.layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        assert self.layers == 12, "the closed-form rollout covers ViT-B (only the last block contributes)"
        self.E = int(sd["visual.proj"].shape[1])
        h = lambda t: t.to(dev, torch.float16).contiguous()
        f = lambda t: t.to(dev, torch.float32).contiguous()
        self.w_patch = h(w.float().reshape(self.D, -1))                     # [D, 3 p p], column = c*p*p + iy*p + ix
        pe = sd["visual.positional_embedding"].float()
        self.pos = f(_interp_pos_emb(pe, self.T) if self.T != 50 else pe)
        self.cls = f(sd["visual.class_embedding"])
        self.ln_pre = (f(sd["visual.ln_pre.weight"]), f(sd["visual.ln_pre.bias"]))
        self.ln_post = (f(sd["visual.ln_post.weight"]), f(sd["visual.ln_post.bias"]))
        self.proj = h(sd["visual.proj"])                                    # [D, E] = B operand of dy = dfeat . proj^T
        self.proj_t = h(sd["visual.proj"].float().t())                      # [E, D] = B operand of feat = y . proj
        self.proj2 = torch.cat([self.proj, self.proj], dim=1).contiguous()          # [W | W] along K for split A operands (head_split)
        self.proj_t2 = torch.cat([self.proj_t, self.proj_t], dim=1).contiguous()
        self.blocks = [_BlockWeights(sd, f"visual.transformer.resblocks.{i}.", self.D, heads, dev, last=(i == self.layers - 1))
                       for i in range(self.layers)]
        # Split operands in the last block and the VJP chain (round 6, default on; SEMABS_HEAD_SPLIT=0: A/B): every fp16 GEMM A operand behind the trunk -
        # the last LayerNorm output feeding K and the CLS query, the CLS row's attention output / LayerNorm / QuickGELU outputs, the logit gradient and the
        # LayerNorm / QuickGELU VJP outputs - is carried as an [hi | lo] pair (K doubled against [W | W]): n or L n rows except the K projection, ~1 % of the
        # flops.  With the peaked softmax of a trained checkpoint (CLS scores up to ~60) the fp16 rounding of that LayerNorm output alone moved the kept
        # softmax row by 2e-3 and the per-tile relevance by 3 - 5e-3 of its maximum (tests/test_gpu_trained_stats.py; budget: test_vit_precision_budget.py).
        self.head_split = os.environ.get("SEMABS_HEAD_SPLIT", "1") == "1"
        self.cls_scores = os.environ.get("SEMABS_CLS_SCORES", "1") == "1"      # A/B: 0 = K projection for all tokens + semabs_attention_cls
        self.chunk = int(chunk_tiles)
        self.max_labels = int(max_labels)
        self._wss = {}
        self._cap = {}           # tiles the workspace of each slot is sized for
        self.delta_residual = DELTA_RESIDUAL
        # Zigzag schedule of the trunk: consecutive kernels walk the token rows in opposite directions (per XCD run), so each starts with the
        # rows its producer wrote last - still in the 256 MiB Infinity Cache - instead of the ones written first (semabs_common.h)
        self.zigzag = os.environ.get("SEMABS_ZIGZAG", "1") == "1"
        # LayerNorm fold (trunk, large batches): ln_2 into out-proj -> c_fc and ln_1 into c_proj -> in_proj (csrc/gemm.hip LNP / LNC).  21 of the 26
        # LayerNorm passes of a scene disappear (each a 1.48 GB fp32 read + 0.74 GB fp16 write at the benchmark batch); the residual GEMMs write the
        # fp16 (x * gamma) copy from their epilogue instead.  SEMABS_LN_FOLD=0 runs the LayerNorm kernels (A/B).
        self.ln_fold = os.environ.get("SEMABS_LN_FOLD", "1") == "1"
        # Row centring of the fold (round 6): the fp16 copy is fp16((x - c) * gamma) with c = the row's mean at its previous LayerNorm.  Released
        # checkpoints carry a per-token DC offset of several sigma of the bulk; un-centred, the copy rounds relative to |mean| instead of |x - mean|
        # (tests/test_gpu_trained_stats.py, test_gpu_gemm.py::test_gemm_layernorm_fold_with_row_dc_offset_and_massive_channels).  SEMABS_LN_CENTER=0: A/B.
        self.ln_center = os.environ.get("SEMABS_LN_CENTER", "1") == "1"
        # precision = "parity" (opt-in; ClipWrapper(..., precision="parity") or SEMABS_QK_SPLIT=1): the QKV GEMM also stores the LOW fp16 halves of q and k
        # and the attention kernel forms the scores from hi + lo pairs (three MFMA products in fp32) - the fp16 rounding of q and k is the largest
        # single term of the maps' deviation from the fp32 reference.  + 1.48 GB written and read per block, one attention workgroup per CU.
        self.qk_split = os.environ.get("SEMABS_QK_SPLIT", "0") == "1"
        self._warned_qk = False
        self.slot = 0          # active workspace (one per HIP stream when tile chunks are pipelined on two streams)

    # ---- workspace ---------------------------------------------------------------------------------
    def _reserve(self, n: int):
        """Size the active workspace for batches of up to n tiles (allocated on first use, re-allocated only to grow: a large `chunk_tiles`
        costs nothing until a batch that large arrives)."""
        if self._cap.get(self.slot, 0) < n:
            self._cap[self.slot] = int(n)
            self._wss.pop(self.slot, None)

    def _workspace(self):
        if self.slot not in self._wss:
            n, T, D, Lm, E = self._cap.get(self.slot) or self.chunk, self.T, self.D, self.max_labels, self.E
            dev = self.dev
            e16 = lambda *s: torch.empty(*s, dtype=torch.float16, device=dev)
            e32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            R = Lm * n
            self._wss[self.slot] = dict(
                x=e32(n * T, D), h=e16(n * T, D), delta=e16(n * T, D), qkv=e16(n * T, 3 * D), att=e16(n * T, D), hid=e16(n * T, 4 * D),
                ln_part=e32(n * T, max(1, D // 256), 2), ln_rowac=e32(n * T, 2), ln_center=e32(n * T), qk_lo=e16(n * T, 2 * D) if self.qk_split else None,
                k32=e32(n * T, D) if not (self.cls_scores and D == 768 and T <= 224) else None, v16=e16(n * T, D), q32=e32(n, D), probs=e32(n, self.H, T), o_cls=e16(n, 2 * D), x1c=e32(n, D),
                h2c=e16(n, 2 * D), fc=e32(n, 4 * D), actc=e16(n, 8 * D), x2c=e32(n, D), yc=e16(n, 2 * D), feat=e32(n, E),
                logits=e32(n, Lm), dfeat=e16(R, 2 * E), scale=e32(R), dy=e32(R, D), dx2=e32(R, D), dx2h=e16(R, 2 * D),
                dact=e32(R, 4 * D), dfc=e16(R, 8 * D), dh2=e32(R, D), g1h=e16(R, 2 * D), u=e32(R, D),
                h_last=e16(n * T, 2 * D) if self.head_split else None,      # the last block's LayerNorm-1 output as [hi | lo] rows (the trunk's `h` is [n T, D])
            )
        return self._wss[self.slot]

    # ---- forward ---------------------------------------------------------------------------------
    def embed(self, patches: torch.Tensor, n: int, defer_ln_pre: bool = False):
        """patches fp16 [n * g*g, 3 p p] -> ws['x'] = ln_pre(cat(cls, conv) + pos)  [n * T, D] fp32.
        defer_ln_pre: leave x un-normalised; trunk() then runs ln_pre and block 0's ln_1 as ONE pass (semabs_layernorm2: bit-identical, one read of x less)."""
        assert n <= self.chunk
        self._reserve(n)
        ws = self._workspace()
        T, D, G = self.T, self.D, self.g * self.g
        x = ws["x"]
        gemm(patches, self.w_patch, x, None, n * G, D, 3 * self.p * self.p, 3 * self.p * self.p, 3 * self.p * self.p, D,
             EPI_ROWMAP, addend=self.pos, rowmap=(G, T, 1))
        _lib.call("semabs_embed_finish", _lib.ptr(x), _lib.ptr(self.cls), _lib.ptr(self.pos), n, T, D, _lib.stream())
        self._ln_pre_pending = bool(defer_ln_pre) and self._fold_taken(n)
        if not self._ln_pre_pending:
            layernorm(x, *self.ln_pre, x, n * T, D, out_f32=True)

    def _fold_taken(self, n: int) -> bool:
        """trunk(n) will take the LayerNorm-fold launch sequence (the one the benchmark batch runs)."""
        M, D = n * self.T, self.D
        return (not self.delta_residual) and self.ln_fold and M >= 2048 and D % 256 == 0

    def trunk(self, n: int):
        """blocks 0 .. layers-2 on ws['x'] in place."""
        ws = self._workspace()
        T, D, H = self.T, self.D, self.H
        M = n * T
        x, h, qkv, att, hid = ws["x"], ws["h"], ws["qkv"], ws["att"], ws["hid"]
        if not self.delta_residual:
            zz = self.zigzag and M >= 2048
            c = [0]

            def step():                                      # direction of the next launch: 0 = forwards, 1 = backwards
                c[0] ^= 1
                return c[0] ^ 1
            if self.ln_fold and M >= 2048 and D % 256 == 0 and D % 128 == 0:
                # h doubles as xg = fp16(x * gamma_next): written by the residual GEMMs' epilogues, read as the next GEMM's A operand
                part, rowac, nt = ws["ln_part"], ws["ln_rowac"], D // 256
                # row centres (LN_CENTER): block 0's ln_1 pass leaves the row means; every producer subtracts the mean its row had at the previous LayerNorm
                # before the fp16 copy, ln_rowstats turns the centred partials into the consumer's pair and the row's new mean
                cen = ws["ln_center"] if self.ln_center else None
                trunk_blocks = self.blocks[:-1]
                d = (lambda: step()) if zz else (lambda: 0)
                qk_lo = ws["qk_lo"] if (self.qk_split and (2 * D) % 256 == 0) else None
                for bi, b in enumerate(trunk_blocks):
                    if bi == 0:                              # ln_1 of the first block follows ln_pre, not a GEMM
                        if getattr(self, "_ln_pre_pending", False):
                            self._ln_pre_pending = False     # ln_pre (in place, fp32) and ln_1 (fp16) in one pass over the rows
                            _lib.call("semabs_layernorm2", _lib.ptr(x), _lib.ptr(self.ln_pre[0]), _lib.ptr(self.ln_pre[1]), _lib.ptr(x), _lib.ptr(b.ln1_w),
                                      _lib.ptr(b.ln1_b), _lib.ptr(h), M, D, 1e-5, (1 + d()) if zz else 0, _lib.ptr(cen), _lib.stream())
                        else:
                            layernorm(x, b.ln1_w, b.ln1_b, h, M, D, order=(1 + d()) if zz else 0, mean_out=cen)
                        if qk_lo is None:
                            gemm(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16, kernel=2 | (d() << 8))
                        else:
                            gemm_ln(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16, reverse=d(), lo=qk_lo, lo_cols=2 * D)
                    else:
                        gemm_ln(h, b.w_in, qkv, b.b_in_f, M, 3 * D, D, D, D, 3 * D, EPI_F16, rowac=rowac, colsum=b.cs_in, reverse=d(), lo=qk_lo, lo_cols=2 * D)
                    if qk_lo is None:
                        _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(att), None, n, T, H, 64, 3 * D, ((1 + d()) << 3) if zz else 0, _lib.stream())
                    else:
                        _lib.call("semabs_attention_split", _lib.ptr(qkv), _lib.ptr(qk_lo), _lib.ptr(att), None, n, T, H, 64, 3 * D, 2 * D,
                                  ((1 + d()) << 3) if zz else 0, _lib.stream())
                    gemm_ln(att, b.w_o, x, b.b_o, M, D, D, D, D, D, EPI_RESID_F32, xg=h, gamma=b.ln2_w, part=part, reverse=d(), center=cen)
                    ln_rowstats(part, M, nt, D, rowac, center=cen, center_out=cen)
                    gemm_ln(h, b.w_fc, hid, b.b_fc_f, M, 4 * D, D, D, D, 4 * D, EPI_GELU_F16, rowac=rowac, colsum=b.cs_fc, reverse=d())
                    if bi + 1 < len(trunk_blocks):
                        gemm_ln(hid, b.w_pr, x, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_RESID_F32, xg=h, gamma=trunk_blocks[bi + 1].ln1_w, part=part, reverse=d(), center=cen)
                        ln_rowstats(part, M, nt, D, rowac, center=cen, center_out=cen)
                    else:                                    # the last block's ln_1 feeds K | V | the CLS query: stays a LayerNorm pass (head)
                        gemm(hid, b.w_pr, x, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_RESID_F32, kernel=2 | (d() << 8))
                return
            if self.qk_split and not self._warned_qk:
                # precision = "parity" needs the persistent QKV kernel (M >= 2048 token rows, D % 256 == 0, LayerNorm fold on); say so instead of silently
                # returning default-precision maps
                import warnings
                warnings.warn(f'precision="parity" (fp16 hi + lo q | k) is not available for this batch ({M} token rows < 2048 or SEMABS_LN_FOLD=0): '
                              "running the default precision", RuntimeWarning, stacklevel=2)
                self._warned_qk = True
            for b in self.blocks[:-1]:
                if not zz:
                    layernorm(x, b.ln1_w, b.ln1_b, h, M, D)
                    gemm(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16)
                    _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(att), None, n, T, H, 64, 3 * D, 0, _lib.stream())
                    gemm(att, b.w_o, x, b.b_o, M, D, D, D, D, D, EPI_RESID_F32)
                    layernorm(x, b.ln2_w, b.ln2_b, h, M, D)
                    gemm(h, b.w_fc, hid, b.b_fc, M, 4 * D, D, D, D, 4 * D, EPI_GELU_F16)
                    gemm(hid, b.w_pr, x, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_RESID_F32)
                    continue
                layernorm(x, b.ln1_w, b.ln1_b, h, M, D, order=1 + step())
                gemm(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16, kernel=2 | (step() << 8))
                _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(att), None, n, T, H, 64, 3 * D, (1 + step()) << 3, _lib.stream())
                gemm(att, b.w_o, x, b.b_o, M, D, D, D, D, D, EPI_RESID_F32, kernel=2 | (step() << 8))
                layernorm(x, b.ln2_w, b.ln2_b, h, M, D, order=1 + step())
                gemm(h, b.w_fc, hid, b.b_fc, M, 4 * D, D, D, D, 4 * D, EPI_GELU_F16, kernel=2 | (step() << 8))
                gemm(hid, b.w_pr, x, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_RESID_F32, kernel=2 | (step() << 8))
            return
        # Residual updates as fp16 deltas: the out-proj / c_proj GEMMs write `delta = A W^T + b` in fp16 (a quarter of the bytes of the fp32
        # read-modify-write) and the following LayerNorm pass, which reads the row anyway, does x += delta before normalising.
        delta = ws["delta"]
        pending = False
        for b in self.blocks[:-1]:
            if pending:
                add_layernorm(x, delta, b.ln1_w, b.ln1_b, h, M, D)
            else:
                layernorm(x, b.ln1_w, b.ln1_b, h, M, D)
            gemm(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16)
            _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(att), None, n, T, H, 64, 3 * D, 0, _lib.stream())
            gemm(att, b.w_o, delta, b.b_o, M, D, D, D, D, D, EPI_F16)
            add_layernorm(x, delta, b.ln2_w, b.ln2_b, h, M, D)
            gemm(h, b.w_fc, hid, b.b_fc, M, 4 * D, D, D, D, 4 * D, EPI_GELU_F16)
            gemm(hid, b.w_pr, delta, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_F16)
            pending = True
        if pending:
            add_layernorm(x, delta, None, None, None, M, D)            # the last block's MLP delta

    def head(self, n: int):
        """last block for the CLS token + ln_post + proj -> ws['feat'] [n, E]; keeps probs / v16 / x1c / fc / x2c.
        head_split: every fp16 A operand is an [hi | lo] row pair of pitch 2 K, multiplied against [W | W] (K doubled)."""
        ws = self._workspace()
        T, D, H, E = self.T, self.D, self.H, self.E
        b = self.blocks[-1]
        x = ws["x"]
        st = _lib.stream()
        sp = 1 if self.head_split else 0
        k2 = 2 if sp else 1                                   # K multiplier of the split operands
        h = ws["h_last"] if sp else ws["h"]
        layernorm(x, b.ln1_w, b.ln1_b, h, n * T, D, out_f32=8 if sp else 0)
        # K and V for every token.  K in fp32: it feeds the kept softmax row directly.  V in fp16 like every other block's: it only enters
        # through averages (the CLS output, itself stored in fp16, and the rollout's 64-long V . u dots), where its rounding is ~6e-5
        # relative - and it is read five times (CLS attention + four label groups of the rollout), so its width is bandwidth
        gemm(h, b.w_in[2 * D:], ws["v16"], b.b_in[2 * D:], n * T, D, D, k2 * D, D, D, EPI_F16)          # the hi halves only (row pitch k2 D)
        # Q for the CLS rows only (row stride T * k2 D)
        gemm(h, b.w_q2 if sp else b.w_in[:D], ws["q32"], b.b_in[:D], n, D, k2 * D, T * k2 * D, k2 * D, D, EPI_F32, split=bool(sp))
        if self.cls_scores and D == 768 and T <= 224:
            # the kept softmax row without the K projection (round 6): s[h, j] = (W_k,h^T q_h) . x_j + q_h . b_k,h - only the CLS query's scores exist in this
            # block, so the [M, D] x [D, D] GEMM with fp32 output (1.1 ms per scene with split rows; 1.48 GB written and read back) is 12 x D values per tile instead
            _lib.call("semabs_cls_scores", _lib.ptr(ws["q32"]), _lib.ptr(b.w_in[D:2 * D]), _lib.ptr(b.b_in[D:2 * D]), _lib.ptr(h), k2 * D, sp, _lib.ptr(ws["probs"]),
                      n, T, D, st)
            _lib.call("semabs_attention_cls", None, None, _lib.ptr(ws["v16"]), _lib.ptr(ws["probs"]), _lib.ptr(ws["o_cls"]), n, T, H, 64, sp, st)
        else:
            gemm(h, b.w_k2 if sp else b.w_in[D:2 * D], ws["k32"], b.b_in[D:2 * D], n * T, D, k2 * D, k2 * D, k2 * D, D, EPI_F32, split=bool(sp))
            _lib.call("semabs_attention_cls", _lib.ptr(ws["q32"]), _lib.ptr(ws["k32"]), _lib.ptr(ws["v16"]), _lib.ptr(ws["probs"]), _lib.ptr(ws["o_cls"]),
                      n, T, H, 64, sp, st)
        _lib.call("semabs_rows_gather", _lib.ptr(x), _lib.ptr(ws["x1c"]), n, D, T * D, 0, st)
        gemm(ws["o_cls"], b.w_o2 if sp else b.w_o, ws["x1c"], b.b_o, n, D, k2 * D, k2 * D, k2 * D, D, EPI_RESID_F32, split=bool(sp))
        layernorm(ws["x1c"], b.ln2_w, b.ln2_b, ws["h2c"], n, D, out_f32=8 if sp else 0)
        gemm(ws["h2c"], b.w_fc2 if sp else b.w_fc, ws["fc"], b.b_fc, n, 4 * D, k2 * D, k2 * D, k2 * D, 4 * D, EPI_F32, split=bool(sp))
        _lib.call("semabs_quickgelu", _lib.ptr(ws["fc"]), _lib.ptr(ws["actc"]), n * 4 * D, 4 * D if sp else 0, st)
        _lib.call("semabs_rows_gather", _lib.ptr(ws["x1c"]), _lib.ptr(ws["x2c"]), n, D, D, 0, st)
        gemm(ws["actc"], b.w_pr2 if sp else b.w_pr, ws["x2c"], b.b_pr, n, D, k2 * 4 * D, k2 * 4 * D, k2 * 4 * D, D, EPI_RESID_F32, split=bool(sp))
        layernorm(ws["x2c"], *self.ln_post, ws["yc"], n, D, out_f32=8 if sp else 0)
        gemm(ws["yc"], self.proj_t2 if sp else self.proj_t, ws["feat"], None, n, E, k2 * D, k2 * D, k2 * D, E, EPI_F32, split=bool(sp))

    def rollout(self, n: int, w_text: torch.Tensor, positive_attn_only: bool, rel_out: torch.Tensor, tile0: int):
        """w_text fp32 [L, E] on the GPU; writes rel_out[:, tile0:tile0+n] (rel_out fp32 [L, N_total, g, g])."""
        ws = self._workspace()
        T, D, H, E = self.T, self.D, self.H, self.E
        L = int(w_text.shape[0])
        assert L <= self.max_labels
        R = L * n
        b = self.blocks[-1]
        st = _lib.stream()
        sp = 1 if self.head_split else 0
        k2 = 2 if sp else 1
        _lib.call("semabs_logit_grad", _lib.ptr(ws["feat"]), _lib.ptr(w_text), n, L, E, _lib.ptr(ws["logits"]),
                  _lib.ptr(ws["dfeat"]), _lib.ptr(ws["scale"]), sp, st)
        gemm(ws["dfeat"], self.proj2 if sp else self.proj, ws["dy"], None, R, D, k2 * E, k2 * E, k2 * E, D, EPI_F32, split=bool(sp))
        _lib.call("semabs_ln_bwd", _lib.ptr(ws["x2c"]), _lib.ptr(self.ln_post[0]), _lib.ptr(ws["dy"]), None,
                  _lib.ptr(ws["dx2"]), _lib.ptr(ws["dx2h"]), R, D, n, D, 1e-5, sp, st)
        gemm(ws["dx2h"], b.w_pr_t2 if sp else b.w_pr_t, ws["dact"], None, R, 4 * D, k2 * D, k2 * D, k2 * D, 4 * D, EPI_F32, split=bool(sp))
        _lib.call("semabs_gelu_bwd", _lib.ptr(ws["dact"]), _lib.ptr(ws["fc"]), _lib.ptr(ws["dfc"]), R, 4 * D, n, sp, st)
        gemm(ws["dfc"], b.w_fc_t2 if sp else b.w_fc_t, ws["dh2"], None, R, D, k2 * 4 * D, k2 * 4 * D, k2 * 4 * D, D, EPI_F32, split=bool(sp))
        _lib.call("semabs_ln_bwd", _lib.ptr(ws["x1c"]), _lib.ptr(b.ln2_w), _lib.ptr(ws["dh2"]), _lib.ptr(ws["dx2"]),
                  None, _lib.ptr(ws["g1h"]), R, D, n, D, 1e-5, sp, st)
        gemm(ws["g1h"], b.w_o_t2 if sp else b.w_o_t, ws["u"], None, R, D, k2 * D, k2 * D, k2 * D, D, EPI_F32, split=bool(sp))
        _lib.call("semabs_rollout", _lib.ptr(ws["probs"]), _lib.ptr(ws["v16"]), _lib.ptr(ws["u"]), _lib.ptr(ws["scale"]),
                  _lib.ptr(rel_out), n, T, H, L, int(positive_attn_only), int(rel_out.shape[1]), int(tile0), st)

    def gradcam_patches(self, patches: torch.Tensor, n: int, w_text: torch.Tensor, positive_attn_only: bool,
                        rel_out: torch.Tensor, tile0: int):
        assert n <= self.chunk
        self.embed(patches, n, defer_ln_pre=os.environ.get("SEMABS_LN2", "1") == "1")
        self.trunk(n)
        self.head(n)
        self.rollout(n, w_text, positive_attn_only, rel_out, tile0)

    def gradcam_tiles(self, tiles: torch.Tensor, w_text: torch.Tensor, positive_attn_only: bool, flip: bool = False):
        """tiles fp32 [n, 3, 224, 224] (GPU) -> (rel [L, n, g, g], logits [n, L], feat [n, E]) — `ClipGradcam.forward`."""
        n = int(tiles.shape[0])
        L = int(w_text.shape[0])
        G, Kp = self.g * self.g, 3 * self.p * self.p
        rel = torch.empty(L, n, self.g, self.g, dtype=torch.float32, device=self.dev)
        logits = torch.empty(n, L, dtype=torch.float32, device=self.dev)
        feat = torch.empty(n, self.E, dtype=torch.float32, device=self.dev)
        patches = torch.empty(min(n, self.chunk) * G, Kp, dtype=torch.float16, device=self.dev)
        for t0 in range(0, n, self.chunk):
            m = min(self.chunk, n - t0)
            tchunk = tiles[t0:t0 + m].contiguous()
            _lib.call("semabs_patchify", _lib.ptr(tchunk), _lib.ptr(patches), m, self.p, int(flip), _lib.stream())
            self.gradcam_patches(patches, m, w_text, positive_attn_only, rel, t0)
            ws = self._workspace()
            logits[t0:t0 + m] = ws["logits"].view(-1)[: m * L].view(m, L)       # the kernel packs rows with stride L
            feat[t0:t0 + m] = ws["feat"][:m]
        return rel, logits, feat


class VisionRolloutDeep:
    """Deep CLIP image tower (ViT-L/14: width 1024, 24 blocks, 16 heads, patch 14 -> 257 tokens) with the TRUE multi-layer rollout of
    `ClipGradcam.interpret` (clip_gradcam.py:70-132): every block i > 10 contributes cam_i = mean_h clamp(grad_i * probs_i), which needs the
    gradient of each label's logit wrt that block's attention probabilities for ALL query rows, i.e. a backward pass through blocks 23 ... 12.

    No autograd: the forward keeps, for the 13 contributing blocks, the residual stream before / between the two sub-layers, the QKV
    projections and the MLP pre-activations; the backward carries one residual gradient per (label, tile) sequence - L * n sequences batched
    as GEMM rows - down the blocks with the LayerNorm / QuickGELU VJP kernels of the ViT-B path, fp16-MFMA GEMMs against transposed weights
    and the attention-backward kernels of csrc/vitl.hip, which also accumulate the rollout update r <- r + r cam_i of the row vector that is
    all the result needs (R[0, 1:]).  The per-sequence gradient is renormalised by a power of two at every block (the chain is linear, cam is
    positively homogeneous) so the fp16 GEMM operands stay in range; the factor is divided back out inside the rollout update.
    Same engine interface as `VisionRollout` (embed / trunk / head / rollout / gradcam_tiles)."""

    def __init__(self, state_dict, chunk_tiles: int = 16, max_labels: int = 16):
        dev = self.dev = _lib.require_gpu()
        sd = state_dict
        w = sd["visual.conv1.weight"]
        self.D, self.p = int(w.shape[0]), int(w.shape[-1])
        self.g = 224 // self.p
        self.T = self.g * self.g + 1
        self.H = self.D // 64
        self.layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        self.first_roll = ROLLOUT_SKIP + 1
        assert self.layers > self.first_roll + 1 and self.T <= 288
        self.E = int(sd["visual.proj"].shape[1])
        h = lambda t: t.to(dev, torch.float16).contiguous()
        f = lambda t: t.to(dev, torch.float32).contiguous()
        self.Kp = 3 * self.p * self.p
        self.Kpad = (self.Kp + 63) // 64 * 64                                # GEMM K granularity: 588 -> 640 zero-padded columns
        wp = torch.zeros(self.D, self.Kpad)
        wp[:, : self.Kp] = w.float().reshape(self.D, -1)
        self.w_patch = h(wp)
        pe = sd["visual.positional_embedding"].float()
        self.pos = f(_interp_pos_emb(pe, self.T) if self.T != 50 else pe)
        self.cls = f(sd["visual.class_embedding"])
        self.ln_pre = (f(sd["visual.ln_pre.weight"]), f(sd["visual.ln_pre.bias"]))
        self.ln_post = (f(sd["visual.ln_post.weight"]), f(sd["visual.ln_post.bias"]))
        self.proj = h(sd["visual.proj"])
        self.proj_t = h(sd["visual.proj"].float().t())
        self.blocks = [_BlockWeights(sd, f"visual.transformer.resblocks.{i}.", self.D, self.H, dev, bwd=(i >= self.first_roll))
                       for i in range(self.layers)]
        self.chunk = int(chunk_tiles)
        self.max_labels = int(max_labels)
        self._wss, self._cap = {}, {}
        self.slot = 0

    def _reserve(self, n: int):
        if self._cap.get(self.slot, 0) < n:
            self._cap[self.slot] = int(n)
            self._wss.pop(self.slot, None)

    def _workspace(self):
        if self.slot not in self._wss:
            n, T, D, Lm, E, H = self._cap.get(self.slot) or self.chunk, self.T, self.D, self.max_labels, self.E, self.H
            dev = self.dev
            e16 = lambda *s: torch.empty(*s, dtype=torch.float16, device=dev)
            e32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            R = Lm * n
            saved = {b: dict(x_in=e32(n * T, D), x_mid=e32(n * T, D), qkv=e16(n * T, 3 * D), fc=e32(n * T, 4 * D), att=e16(n * T, D),
                             fstats=e32(n * H * T, 2))                       # attention output + softmax row statistics: the backward rebuilds P from them
                     for b in range(self.first_roll, self.layers)}
            self._wss[self.slot] = dict(
                x=e32(n * T, D), h=e16(n * T, D), qkv=e16(n * T, 3 * D), att=e16(n * T, D), hid=e16(n * T, 4 * D), saved=saved,
                xcls=e32(n, D), yc=e16(n, D), feat=e32(n, E), logits=e32(n, Lm), dfeat=e16(R, E), scale=e32(R), dy=e32(R, D), dxc=e32(R, D),
                g32=e32(R * T, D), g16=e16(R * T, D), dact=e32(R * T, 4 * D), dfc=e16(R * T, 4 * D), dh=e32(R * T, D), gmid32=e32(R * T, D),
                gmid16=e16(R * T, D), dO=e16(R * T, D), dqkv=e16(R * T, 3 * D), stats=e32(R * H * T, 4), rvec=e32(R, T), cacc=e32(R, T),
                gscale=e32(R))
        return self._wss[self.slot]

    # ---- forward ---------------------------------------------------------------------------------
    def embed(self, patches: torch.Tensor, n: int):
        assert n <= self.chunk
        self._reserve(n)
        ws = self._workspace()
        T, D, G = self.T, self.D, self.g * self.g
        x = ws["x"]
        if self.Kpad != self.Kp:                                             # zero-padded copy (data movement only): rows become 16-byte aligned
            pp = torch.zeros(n * G, self.Kpad, dtype=torch.float16, device=self.dev)
            pp[:, : self.Kp] = patches[: n * G]
            patches = pp
        gemm(patches, self.w_patch, x, None, n * G, D, self.Kpad, self.Kpad, self.Kpad, D, EPI_ROWMAP, addend=self.pos, rowmap=(G, T, 1))
        _lib.call("semabs_embed_finish", _lib.ptr(x), _lib.ptr(self.cls), _lib.ptr(self.pos), n, T, D, _lib.stream())
        layernorm(x, *self.ln_pre, x, n * T, D, out_f32=True)

    def trunk(self, n: int):
        ws = self._workspace()
        T, D, H = self.T, self.D, self.H
        M = n * T
        x, h, att, hid = ws["x"], ws["h"], ws["att"], ws["hid"]
        st = _lib.stream()
        for i, b in enumerate(self.blocks):
            sv = ws["saved"].get(i)
            qkv = ws["qkv"] if sv is None else sv["qkv"]
            att = ws["att"] if sv is None else sv["att"]
            if sv is not None:
                sv["x_in"][:M].copy_(x[:M])
            layernorm(x, b.ln1_w, b.ln1_b, h, M, D)
            gemm(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16)
            _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(att), None if sv is None else _lib.ptr(sv["fstats"]), n, T, H, 64, 3 * D, 0, st)
            gemm(att, b.w_o, x, b.b_o, M, D, D, D, D, D, EPI_RESID_F32)
            if sv is not None:
                sv["x_mid"][:M].copy_(x[:M])
            layernorm(x, b.ln2_w, b.ln2_b, h, M, D)
            if sv is None:
                gemm(h, b.w_fc, hid, b.b_fc, M, 4 * D, D, D, D, 4 * D, EPI_GELU_F16)
            else:                                                            # keep the pre-activation: the QuickGELU VJP needs it
                gemm(h, b.w_fc, sv["fc"], b.b_fc, M, 4 * D, D, D, D, 4 * D, EPI_F32)
                _lib.call("semabs_quickgelu", _lib.ptr(sv["fc"]), _lib.ptr(hid), M * 4 * D, 0, st)
            gemm(hid, b.w_pr, x, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_RESID_F32)

    def head(self, n: int):
        ws = self._workspace()
        T, D, E = self.T, self.D, self.E
        _lib.call("semabs_rows_gather", _lib.ptr(ws["x"]), _lib.ptr(ws["xcls"]), n, D, T * D, 0, _lib.stream())
        layernorm(ws["xcls"], *self.ln_post, ws["yc"], n, D)
        gemm(ws["yc"], self.proj_t, ws["feat"], None, n, E, D, D, D, E, EPI_F32)

    # ---- backward + rollout ------------------------------------------------------------------------
    def rollout(self, n: int, w_text: torch.Tensor, positive_attn_only: bool, rel_out: torch.Tensor, tile0: int):
        ws = self._workspace()
        T, D, H, E = self.T, self.D, self.H, self.E
        L = int(w_text.shape[0])
        assert L <= self.max_labels
        R = L * n
        M = R * T
        st = _lib.stream()
        _lib.call("semabs_logit_grad", _lib.ptr(ws["feat"]), _lib.ptr(w_text), n, L, E, _lib.ptr(ws["logits"]), _lib.ptr(ws["dfeat"]),
                  _lib.ptr(ws["scale"]), 0, st)
        gemm(ws["dfeat"], self.proj, ws["dy"], None, R, D, E, E, E, D, EPI_F32)
        _lib.call("semabs_ln_bwd", _lib.ptr(ws["xcls"]), _lib.ptr(self.ln_post[0]), _lib.ptr(ws["dy"]), None, _lib.ptr(ws["dxc"]), None,
                  R, D, n, D, 1e-5, 0, st)
        g32, g16, gscale, rvec, cacc = ws["g32"], ws["g16"], ws["gscale"], ws["rvec"], ws["cacc"]
        g32[:M].zero_()
        g32[:M].view(R, T, D)[:, 0, :] = ws["dxc"][:R]                       # only the class token of the last block reaches the feature
        gscale[:R].copy_(ws["scale"][:R])
        _lib.call("semabs_seq_rescale", _lib.ptr(g32), _lib.ptr(g16), _lib.ptr(gscale), R, T * D, st)
        rvec[:R].zero_()
        rvec[:R, 0] = 1.0                                                    # row 0 of R = I
        cacc[:R].zero_()
        NT = n * T
        for i in range(self.layers - 1, self.first_roll - 1, -1):
            b, sv = self.blocks[i], ws["saved"][i]
            # MLP sub-layer: g_mid = g_out + LN2^T( W_fc^T( gelu'(fc) * (W_pr^T g_out) ) )
            if M >= 2048 and FUSE_GELU_BWD:                      # the QuickGELU derivative in the GEMM's epilogue: no fp32 dact written and re-read
                _lib.call("semabs_quickgelu_grad", _lib.ptr(sv["fc"]), _lib.ptr(ws["dact"]), NT * 4 * D, st)      # derivative table, shared by the labels (dact = scratch)
                gemm(g16, b.w_pr_t, ws["dfc"], None, M, 4 * D, D, D, D, 4 * D, EPI_GELUBWD, addend=ws["dact"], rowmap=(NT, 1, 0))
            else:
                gemm(g16, b.w_pr_t, ws["dact"], None, M, 4 * D, D, D, D, 4 * D, EPI_F32)
                _lib.call("semabs_gelu_bwd", _lib.ptr(ws["dact"]), _lib.ptr(sv["fc"]), _lib.ptr(ws["dfc"]), M, 4 * D, NT, 0, st)
            gemm(ws["dfc"], b.w_fc_t, ws["dh"], None, M, D, 4 * D, 4 * D, 4 * D, D, EPI_F32)
            _lib.call("semabs_ln_bwd", _lib.ptr(sv["x_mid"]), _lib.ptr(b.ln2_w), _lib.ptr(ws["dh"]), _lib.ptr(g32), _lib.ptr(ws["gmid32"]),
                      _lib.ptr(ws["gmid16"]), M, D, NT, D, 1e-5, 0, st)
            # attention sub-layer: dO = g_mid W_o, then (dQ | dK | dV) and this block's rollout update
            gemm(ws["gmid16"], b.w_o_t, ws["dO"], None, M, D, D, D, D, D, EPI_F16)
            last = i == self.first_roll
            _lib.call("semabs_attention_bwd", _lib.ptr(sv["qkv"]), _lib.ptr(sv["att"]), _lib.ptr(sv["fstats"]), _lib.ptr(ws["dO"]), _lib.ptr(rvec),
                      _lib.ptr(gscale), _lib.ptr(cacc), _lib.ptr(ws["stats"]), None if last else _lib.ptr(ws["dqkv"]), n, L, T, H, 64,
                      int(positive_attn_only), st)
            _lib.call("semabs_rollout_step", _lib.ptr(rvec), _lib.ptr(cacc), R * T, st)
            if not last:
                gemm(ws["dqkv"], b.w_in_t, ws["dh"], None, M, D, 3 * D, 3 * D, 3 * D, D, EPI_F32)
                _lib.call("semabs_ln_bwd", _lib.ptr(sv["x_in"]), _lib.ptr(b.ln1_w), _lib.ptr(ws["dh"]), _lib.ptr(ws["gmid32"]), _lib.ptr(g32),
                          _lib.ptr(g16), M, D, NT, D, 1e-5, 0, st)
                _lib.call("semabs_seq_rescale", _lib.ptr(g32), _lib.ptr(g16), _lib.ptr(gscale), R, T * D, st)
        rel_out[:L, tile0:tile0 + n] = rvec[:R].view(L, n, T)[:, :, 1:].reshape(L, n, self.g, self.g)

    def gradcam_patches(self, patches, n, w_text, positive_attn_only, rel_out, tile0):
        assert n <= self.chunk
        self.embed(patches, n)
        self.trunk(n)
        self.head(n)
        self.rollout(n, w_text, positive_attn_only, rel_out, tile0)

    gradcam_tiles = VisionRollout.gradcam_tiles


class TextEncoder:
    """CLIP text tower on the same kernels (model_explainability.py:469-482) -> zero-shot weights
    (clip_gradcam.py:12-27: per-template L2 normalise, mean over templates, not re-normalised)."""

    def __init__(self, state_dict, heads: int | None = None):
        dev = self.dev = _lib.require_gpu()
        sd = state_dict
        f = lambda t: t.to(dev, torch.float32).contiguous()
        self.emb = f(sd["token_embedding.weight"])
        self.pos = f(sd["positional_embedding"])
        self.D = int(self.emb.shape[1])
        self.ctx = int(self.pos.shape[0])
        self.H = int(heads) if heads else self.D // 64              # transformer_heads = transformer_width // 64 (model_explainability.py:593)
        self.layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
        self.blocks = [_BlockWeights(sd, f"transformer.resblocks.{i}.", self.D, self.H, dev) for i in range(self.layers)]
        self.ln_final = (f(sd["ln_final.weight"]), f(sd["ln_final.bias"]))
        self.proj_t = sd["text_projection"].float().t().to(dev, torch.float16).contiguous()     # [E, D]
        self.E = int(self.proj_t.shape[0])

    def zeroshot_weights(self, tokens: torch.Tensor, n_classes: int, n_templates: int) -> torch.Tensor:
        """tokens int64 [n_classes * n_templates, 77] (class-major) -> fp32 [n_classes, E] on the GPU."""
        assert int(tokens.shape[0]) == n_classes * n_templates
        e = self.encode(tokens)
        w = torch.empty(n_classes, self.E, dtype=torch.float32, device=self.dev)
        _lib.call("semabs_text_finish", _lib.ptr(e), _lib.ptr(w), n_classes, n_templates, self.E, _lib.stream())
        return w

    def encode(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens int64 [B, 77] -> fp32 [B, E] on the GPU (`CLIP.encode_text`)."""
        dev, D, T, H, E = self.dev, self.D, self.ctx, self.H, self.E
        B = int(tokens.shape[0])
        assert tokens.shape[1] == T
        tok = tokens if (tokens.is_cuda and tokens.dtype == torch.int64 and tokens.is_contiguous()) else tokens.to(dev, torch.int64).contiguous()
        M = B * T
        x = torch.empty(M, D, dtype=torch.float32, device=dev)
        h = torch.empty(M, D, dtype=torch.float16, device=dev)
        qkv = torch.empty(M, 3 * D, dtype=torch.float16, device=dev)
        att = torch.empty(M, D, dtype=torch.float16, device=dev)
        hid = torch.empty(M, 4 * D, dtype=torch.float16, device=dev)
        st = _lib.stream()
        _lib.call("semabs_gather_text", _lib.ptr(tok), _lib.ptr(self.emb), _lib.ptr(self.pos), _lib.ptr(x), B, T, D, st)
        for b in self.blocks:
            layernorm(x, b.ln1_w, b.ln1_b, h, M, D)
            gemm(h, b.w_in, qkv, b.b_in, M, 3 * D, D, D, D, 3 * D, EPI_F16)
            _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(att), None, B, T, H, 64, 3 * D, 1, st)
            gemm(att, b.w_o, x, b.b_o, M, D, D, D, D, D, EPI_RESID_F32)
            layernorm(x, b.ln2_w, b.ln2_b, h, M, D)
            gemm(h, b.w_fc, hid, b.b_fc, M, 4 * D, D, D, D, 4 * D, EPI_GELU_F16)
            gemm(hid, b.w_pr, x, b.b_pr, M, D, 4 * D, 4 * D, 4 * D, D, EPI_RESID_F32)
        # EOT rows (the highest token id in each sequence) -> ln_final -> text_projection
        xe = torch.empty(B, D, dtype=torch.float32, device=dev)
        _lib.call("semabs_eot_rows_gather", _lib.ptr(tok), _lib.ptr(x), _lib.ptr(xe), B, T, D, st)
        ye = torch.empty(B, D, dtype=torch.float16, device=dev)
        layernorm(xe, *self.ln_final, ye, B, D)
        e = torch.empty(B, E, dtype=torch.float32, device=dev)
        gemm(ye, self.proj_t, e, None, B, E, D, D, D, E, EPI_F32)
        return e
