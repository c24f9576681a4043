"""ORACLE (test infrastructure, not product code) — CPU restatement of the multi-scale CLIP relevancy extractor.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Restates, in plain torch-CPU fp32 / numpy, what the reference computes on its CPU path:

  saliency_configs                      CLIP/clip/__init__.py:19-41
  tile geometry + counts                CLIP/clip/__init__.py:238-282   (create_tiles)
  hooked ViT forward                    CLIP/clip/model_explainability.py:324-355, 232-255
                                        CLIP/clip/auxiliary.py:24-38 (pos-emb quirk), :117-347 (MHA)
  zero-shot text weights                CLIP/clip/clip_gradcam.py:12-27, model_explainability.py:469-482
  logits + attention x gradient rollout CLIP/clip/clip_gradcam.py:58-132
  flip / upsample / fp16 accumulate     CLIP/clip/__init__.py:135-236

The rollout is written in closed form instead of L autograd passes.  For ViT-B the loop at
clip_gradcam.py:85-87 keeps only block 11 (i <= num_layers=10 is skipped), R = I + cam and the output is
cam[CLS, 1:].  Only the CLS query row of block 11 reaches the image feature, so with
g1 = d logit / d x1[CLS] (x1 = residual stream after block-11 attention):
    grad[l, n, h, 0, j] = V[n, h, j, :] . (g1[l, n] @ W_o)[h*dh:(h+1)*dh]
    rel[l, n, j-1]      = mean_h clamp(A[n, h, 0, j] * grad, min=0 if positive_attn_only)
Pinned against the reference's autograd result (tests/golden, G3/G4/G6).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .preprocess import preprocess_tile

# ------------------------------------------------------------------------------------------------
# a1: saliency configs (CLIP/clip/__init__.py:19-41)
# ------------------------------------------------------------------------------------------------
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


# ------------------------------------------------------------------------------------------------
# a2: tile geometry (CLIP/clip/__init__.py:254-274).  Rows are named x and columns y there.
# ------------------------------------------------------------------------------------------------
def tile_table(H: int, W: int, n_images: int, cropping_augmentations) -> np.ndarray:
    """int32 [N, 4] rows (image, row0, col0, tile_size) in the reference's append order."""
    rows = []
    for im in range(n_images):
        for aug in cropping_augmentations:
            ts, stride = int(aug["tile_size"]), int(aug["stride"])
            for y in np.arange(0, W - ts + 1, stride):
                if y >= H:
                    continue
                for x in np.arange(0, H - ts + 1, stride):
                    if x >= W:
                        continue
                    rows.append((im, int(x), int(y), ts))
    return np.asarray(rows, dtype=np.int32).reshape(-1, 4)


def tile_counts(H: int, W: int, table: np.ndarray, tile_sizes: Sequence[int] = ()) -> Dict[int, np.ndarray]:
    """fp32 [H, W] per tile size, 1e-5 + number of covering tiles (over all images), insertion-ordered.
    `tile_sizes`: every crop_aug's tile size in config order — the reference creates a canvas for each of them
    (CLIP/clip/__init__.py:249-253), including a scale that ends up with no tile (it then contributes 0 / 1e-5 = 0 and
    still counts in the mean over scales)."""
    counts: Dict[int, np.ndarray] = {int(ts): np.zeros((H, W), np.float32) + np.float32(1e-5) for ts in tile_sizes}
    for im, x, y, ts in table:
        c = counts.setdefault(int(ts), np.zeros((H, W), np.float32) + np.float32(1e-5))
        c[x:x + ts, y:y + ts] += 1
    return counts


def make_tile_images(images: Sequence[np.ndarray], table: np.ndarray) -> torch.Tensor:
    """fp32 [N, 3, 224, 224]: crop -> PIL-exact bicubic 224 -> /255 -> normalise."""
    out = np.empty((len(table), 3, 224, 224), np.float32)
    for i, (im, x, y, ts) in enumerate(table):
        out[i] = preprocess_tile(images[im][x:x + ts, y:y + ts])
    return torch.from_numpy(out)


# ------------------------------------------------------------------------------------------------
# a5/a6: hooked ViT forward
# ------------------------------------------------------------------------------------------------
def interpolate_positional_emb(pe: torch.Tensor, T: int) -> torch.Tensor:
    """The hard-coded-50 "interpolation" (auxiliary.py:24-38); taken whenever T != 50."""
    out = torch.zeros(T, pe.shape[1], dtype=pe.dtype)
    for i in range(T):
        i3 = float(i) / (T / 50)
        i1, i2 = math.floor(i3), math.ceil(i3)
        if i2 < len(pe):
            out[i] = torch.lerp(pe[i1], pe[i2], i3 - i1)
        else:
            out[i] = pe[-1]
    return out


def _ln(x, w, b):
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5)


def _block(sd, pre, x, heads, attn_mask=None, want=None):
    """One ResidualAttentionBlock on x [n, T, D] (batch-first; the reference uses LND, same math).
    want: dict filled with probs [n, H, T, T], v [n, H, T, dh], x1 (after attention residual)."""
    n, T, D = x.shape
    dh = D // heads
    h = _ln(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = F.linear(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q * (float(dh) ** -0.5)
    q = q.view(n, T, heads, dh).transpose(1, 2)
    k = k.view(n, T, heads, dh).transpose(1, 2)
    v = v.view(n, T, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if attn_mask is not None:
        s = s + attn_mask
    p = F.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(n, T, D)
    x1 = x + F.linear(o, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])
    h2 = _ln(x1, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    fc = F.linear(h2, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"])
    act = fc * torch.sigmoid(1.702 * fc)
    x2 = x1 + F.linear(act, sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])
    if want is not None:
        want.update(probs=p, v=v, x1=x1, h2=h2, fc=fc, x2=x2)
    return x2


def vit_embed(sd, tiles: torch.Tensor) -> torch.Tensor:
    """patch conv + class token + positional embedding (with the quirk) + ln_pre -> [n, T, D]."""
    w = sd["visual.conv1.weight"]
    p = w.shape[-1]
    x = F.conv2d(tiles, w, stride=p)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"] + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1)
    T = x.shape[1]
    pe = sd["visual.positional_embedding"]
    if T != 50:
        x = x + interpolate_positional_emb(pe, T)
    else:
        x = x + pe
    return _ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])


def vit_forward(sd, tiles: torch.Tensor, heads: int = 12, layers: int = 12):
    """-> (image feature [n, E], last-block intermediates dict)."""
    x = vit_embed(sd, tiles)
    last = {}
    for i in range(layers):
        x = _block(sd, f"visual.transformer.resblocks.{i}.", x, heads, want=last if i == layers - 1 else None)
    y = _ln(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    feat = y @ sd["visual.proj"]
    last["x_final_cls"] = x[:, 0, :]
    last["ln_post_out"] = y
    return feat, last


# ------------------------------------------------------------------------------------------------
# a4: text tower -> zero-shot weights
# ------------------------------------------------------------------------------------------------
def encode_text(sd, tokens: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """tokens int64 [B, 77] -> [B, E] (model_explainability.py:469-482)."""
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    L = x.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(layers):
        x = _block(sd, f"transformer.resblocks.{i}.", x, heads, attn_mask=mask)
    x = _ln(x, sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd["text_projection"]


def zeroshot_weights(sd, tokens: torch.Tensor, n_classes: int, n_templates: int) -> torch.Tensor:
    """-> [E, n_classes]: per-template L2 normalise, mean over templates, NOT re-normalised
    (clip_gradcam.py:22-27).  `tokens` is class-major: row c * n_templates + t."""
    e = encode_text(sd, tokens).view(n_classes, n_templates, -1)
    e = e / e.norm(dim=-1, keepdim=True)
    return e.mean(dim=1).T


# ------------------------------------------------------------------------------------------------
# a7/a8: logits + closed-form last-block attention x gradient
# ------------------------------------------------------------------------------------------------
def _ln_vjp(x, w, gy, eps=1e-5):
    """VJP of y = LN(x) * w + b wrt x, rows independent."""
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    rstd = (var + eps).rsqrt()
    xh = (x - mu) * rstd
    g = gy * w
    return rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))


def gradcam_tiles(sd, tiles: torch.Tensor, w_text: torch.Tensor, positive_attn_only: bool,
                  heads: int = 12, layers: int = 12):
    """tiles fp32 [n, 3, 224, 224], w_text [E, L] -> (rel [L, n, g, g], logits [n, L])."""
    if len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")]) > 12:
        return gradcam_tiles_deep(sd, tiles, w_text, positive_attn_only)          # ViT-L/14: several blocks enter the rollout
    pre = f"visual.transformer.resblocks.{layers - 1}."
    feat, last = vit_forward(sd, tiles, heads, layers)
    n, E = feat.shape
    L = w_text.shape[1]
    nrm = feat.norm(dim=-1, keepdim=True)
    fh = feat / nrm
    logits = 100.0 * fh @ w_text                                             # clip_gradcam.py:63-67
    # d logit_l / d feat  [L, n, E]
    wl = w_text.T[:, None, :]                                                  # [L, 1, E]
    dfeat = 100.0 * (wl - fh[None] * (fh[None] * wl).sum(-1, keepdim=True)) / nrm[None]
    # feat = ln_post(x2c) @ proj
    dy = dfeat @ sd["visual.proj"].T                                          # [L, n, D]
    x2c = last["x_final_cls"]
    dx2 = _ln_vjp(x2c[None], sd["visual.ln_post.weight"], dy)
    # x2c = x1c + c_proj(gelu(c_fc(ln_2(x1c))))
    x1c = last["x1"][:, 0, :]
    fc = last["fc"][:, 0, :]
    dact = dx2 @ sd[pre + "mlp.c_proj.weight"]                                # [L, n, 4D]
    sg = torch.sigmoid(1.702 * fc)
    dfc = dact * (sg * (1 + 1.702 * fc * (1 - sg)))[None]
    dh2 = dfc @ sd[pre + "mlp.c_fc.weight"]
    g1 = dx2 + _ln_vjp(x1c[None], sd[pre + "ln_2.weight"], dh2)               # d logit / d x1[CLS]
    u = g1 @ sd[pre + "attn.out_proj.weight"]                                  # [L, n, D] = W_o^T g1
    D = u.shape[-1]
    dh = D // heads
    u = u.view(L, n, heads, dh)
    v = last["v"]                                                              # [n, H, T, dh]
    grad = torch.einsum("nhjd,lnhd->lnhj", v, u)                               # grad wrt probs[n,h,0,j]
    cam = grad * last["probs"][:, :, 0, :][None]
    if positive_attn_only:
        cam = cam.clamp(min=0)
    cam = cam.mean(dim=2)                                                      # [L, n, T]
    g = int(round(math.sqrt(cam.shape[-1] - 1)))
    return cam[:, :, 1:].reshape(L, n, g, g), logits


def gradcam_tiles_deep(sd, tiles: torch.Tensor, w_text: torch.Tensor, positive_attn_only: bool, num_layers: int = 10):
    """Multi-layer rollout for towers deeper than ViT-B (ViT-L/14: 24 blocks, heads = width / 64) - ClipGradcam.interpret restated with
    torch autograd like the reference itself (clip_gradcam.py:70-132): per label, the gradient of sum_n logit[n, l] wrt the attention
    probabilities of every block i > num_layers; cam = mean_h clamp(grad * probs); R <- R + cam R; returns R[:, :, 0, 1:].
    tiles [n, 3, 224, 224], w_text [E, L] -> (rel [L, n, g, g], logits [n, L])."""
    D = sd["visual.conv1.weight"].shape[0]
    heads = D // 64
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    with torch.enable_grad():
        x = vit_embed(sd, tiles)
        probs = []
        for i in range(layers):
            want = {}
            pre = f"visual.transformer.resblocks.{i}."
            if i <= num_layers:
                x = _block(sd, pre, x, heads)
                continue
            # same block, with the softmax output exposed as a leaf-like node of the graph
            n, T, _ = x.shape
            dh = D // heads
            h = _ln(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
            q, k, v = F.linear(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"]).chunk(3, dim=-1)
            q = (q * (float(dh) ** -0.5)).view(n, T, heads, dh).transpose(1, 2)
            k = k.view(n, T, heads, dh).transpose(1, 2)
            v = v.view(n, T, heads, dh).transpose(1, 2)
            p = F.softmax(q @ k.transpose(-1, -2), dim=-1)
            if not p.requires_grad:
                p.requires_grad_(True)
            probs.append(p)
            o = (p @ v).transpose(1, 2).reshape(n, T, D)
            x1 = x + F.linear(o, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])
            h2 = _ln(x1, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
            fc = F.linear(h2, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"])
            x = x1 + F.linear(fc * torch.sigmoid(1.702 * fc), sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])
        feat = _ln(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
        fh = feat / feat.norm(dim=-1, keepdim=True)
        logits = 100.0 * fh @ w_text
        n, L, T = logits.shape[0], logits.shape[1], probs[0].shape[-1]
        R = torch.eye(T)[None, None].repeat(L, n, 1, 1)
        one_hot = logits.sum(dim=0)
        for p in probs:
            grad = torch.stack([torch.autograd.grad(one_hot[l], [p], retain_graph=True)[0].detach() for l in range(L)])      # [L, n, H, T, T]
            cam = grad * p.detach()[None]
            if positive_attn_only:
                cam = cam.clamp(min=0)
            cam = cam.mean(dim=2)
            R = R + torch.matmul(cam, R)
    g = int(round(math.sqrt(T - 1)))
    return R[:, :, 0, 1:].reshape(L, n, g, g), logits.detach()


# ------------------------------------------------------------------------------------------------
# a9: flip pass, un-flip average, bilinear upsample, fp16 canvases, count-normalise, mean over scales
# ------------------------------------------------------------------------------------------------
def aggregate(rel: torch.Tensor, table: np.ndarray, H: int, W: int,
              tile_interpolate_batch_size: int = 32, tile_sizes: Sequence[int] = ()) -> torch.Tensor:
    """rel fp32 [L, N, g, g] (already flip-averaged) -> fp32 [L, H, W]  (__init__.py:205-236)."""
    counts = tile_counts(H, W, table, tile_sizes)
    L = rel.shape[0]
    outputs = {k: torch.zeros(L, H, W).half() for k in counts}
    sizes = table[:, 3]
    for ts in np.unique(sizes):
        sel = np.nonzero(sizes == ts)[0]
        cur = rel[:, sel]
        for b0 in range(0, len(sel), tile_interpolate_batch_size):
            up = F.interpolate(cur[:, b0:b0 + tile_interpolate_batch_size], size=int(ts),
                               mode="bilinear", align_corners=False)
            for j, ti in enumerate(sel[b0:b0 + tile_interpolate_batch_size]):
                _, x, y, _ = table[ti]
                outputs[int(ts)][:, x:x + ts, y:y + ts] += up[:, j]
    return sum(o.float() / torch.from_numpy(c) for o, c in zip(outputs.values(), counts.values())) / len(counts)


def relevancy_maps(sd, images: Sequence[np.ndarray], w_text: torch.Tensor, cropping_augmentations,
                   horizontal_flipping: bool, positive_attn_only: bool, tile_batch_size: int = 32,
                   heads: int = 12, layers: int = 12, return_tiles: bool = False, **_ignored):
    """images: list of uint8 [H, W, 3] (original first, then the already-jittered augmentations)
    -> fp32 [L, H, W]   (get_clip_saliency_convolve, __init__.py:135-236)."""
    H, W = images[0].shape[:2]
    table = tile_table(H, W, len(images), cropping_augmentations)
    tiles = make_tile_images(images, table)

    def run(t):
        outs = [gradcam_tiles(sd, t[i:i + tile_batch_size], w_text, positive_attn_only, heads, layers)[0]
                for i in range(0, len(t), tile_batch_size)]
        return torch.cat(outs, dim=1)

    rel = run(tiles)
    if horizontal_flipping:
        rel_f = run(torch.flip(tiles, dims=[-1]))
        rel = (rel + torch.flip(rel_f, dims=[-1])) / 2
    out = aggregate(rel, table, H, W, tile_sizes=[a["tile_size"] for a in cropping_augmentations])
    return (out, rel, table) if return_tiles else out
