"""Seeded synthetic weights for the two networks on the path.

There is no network here and the released checkpoints (OpenAI CLIP ``ViT-B-32.pt`` / ``ViT-B-16.pt``,
``ovssc.pth``) cannot be fetched, so every run uses random-init weights of the exact released
architectures, generated from a seed with numpy's PCG64 so that the golden-vector generator (which
loads them into the *reference* modules), the CPU oracle and the HIP path all see the same numbers
without a multi-hundred-MB fixture.

Key names follow the checkpoints the reference loads (`CLIP/clip/model_explainability.py:530-602`
for CLIP, `utils.py:276-290` for SemAbs3D) so a real state_dict can be dropped in unchanged.

`build_model` converts Conv/Linear/MHA/proj tensors to fp16 and (on CPU) back to fp32
(`model_explainability.py:501-527`, `clip_explainability.py:165-169`), so those tensors are
fp16-representable; LayerNorm parameters, class/positional/token embeddings stay fp32.
"""
from __future__ import annotations

import numpy as np
import torch

CLIP_ARCHS = {
    # name: (patch, vision_width, vision_layers, vision_heads, embed_dim, text_width, text_heads, text_layers)
    "ViT-B/32": dict(patch=32, width=768, layers=12, heads=12, embed=512, twidth=512, theads=8, tlayers=12),
    "ViT-B/16": dict(patch=16, width=768, layers=12, heads=12, embed=512, twidth=512, theads=8, tlayers=12),
    # clip_gradcam.py:51-56 also lists ViT-L/14: 24 blocks of which 13 (i > 10) enter the rollout
    "ViT-L/14": dict(patch=14, width=1024, layers=24, heads=16, embed=768, twidth=768, theads=12, tlayers=12),
}
CONTEXT_LENGTH = 77
VOCAB_SIZE = 49408
IMAGE_RES = 224


def _f16(x: np.ndarray) -> np.ndarray:
    """Round to the nearest fp16 value, returned as fp32 (what `convert_weights` + `.float()` leaves)."""
    return x.astype(np.float16).astype(np.float32)


def _block(rng, prefix, width, sd, attn_std, proj_std, fc_std, sharpen, half=True):
    r = _f16 if half else (lambda a: a.astype(np.float32))
    n = lambda *s: rng.standard_normal(s, dtype=np.float32)
    w_in = n(3 * width, width) * attn_std
    # sharpen q/k so the softmax is not near-uniform (random CLIP init gives almost flat attention)
    w_in[: 2 * width] *= sharpen
    sd[prefix + "attn.in_proj_weight"] = r(w_in)
    sd[prefix + "attn.in_proj_bias"] = r(n(3 * width) * 0.02)
    sd[prefix + "attn.out_proj.weight"] = r(n(width, width) * proj_std)
    sd[prefix + "attn.out_proj.bias"] = r(n(width) * 0.02)
    sd[prefix + "ln_1.weight"] = 1.0 + 0.1 * n(width)
    sd[prefix + "ln_1.bias"] = 0.05 * n(width)
    sd[prefix + "mlp.c_fc.weight"] = r(n(4 * width, width) * fc_std)
    sd[prefix + "mlp.c_fc.bias"] = r(n(4 * width) * 0.02)
    sd[prefix + "mlp.c_proj.weight"] = r(n(width, 4 * width) * proj_std)
    sd[prefix + "mlp.c_proj.bias"] = r(n(width) * 0.02)
    sd[prefix + "ln_2.weight"] = 1.0 + 0.1 * n(width)
    sd[prefix + "ln_2.bias"] = 0.05 * n(width)


def _trained_tower(sd: dict, prefix: str, ln_in: str, width: int, layers: int, rng, qk_gain, half=True) -> list:
    """In-place edit of one transformer tower towards the activation statistics of a TRAINED CLIP checkpoint (none can be fetched here).
    What released ViT-B/32, B/16 checkpoints show in their residual stream and random-init weights do not:

    * massive-activation channels: three fixed channels carry |x| 50 - 150 x the median of the others from the first blocks on - here through the gains
      and offsets of the tower's input LayerNorm (`ln_in`, vision tower only) and the c_proj rows / biases of blocks 1 - 3; the later LayerNorm gains on those
      channels are small (0.05 - 0.5), as in trained models, so that they dominate the row variance without drowning the projections;
    * a per-row DC offset: every token's channels share an offset of several sigma of the bulk, different per token (a rank-one `1 v^T` component in
      the out_proj / c_proj weights of blocks 0 - 4 plus a constant in their biases) - |mean| >> spread of the bulk is what an un-centred fp16
      copy of the row loses bits to;
    * peaked attention: q / k rows scaled per block (`qk_gain[i]`) so the scaled scores have sigma 2 - 4 in the trunk and 6 - 10 in the last block.
    Returns the massive channel indices."""
    r = _f16 if half else (lambda a: a.astype(np.float32))
    t = lambda k: sd[prefix + k]
    massive = sorted(int(c) for c in rng.choice(width, size=3, replace=False))
    signs = np.array([1.0, -1.0, 1.0], np.float32)
    if ln_in is not None:
        sd[ln_in + ".weight"][massive] = np.array([10.0, 16.0, 7.0], np.float32)
        sd[ln_in + ".bias"][massive] = signs * np.array([35.0, 50.0, 24.0], np.float32)
    ones = np.ones(width, np.float32)
    for i in range(layers):
        pre = f"transformer.resblocks.{i}."
        if 1 <= i <= 3:
            w = t(pre + "mlp.c_proj.weight")
            w[massive] *= 24.0
            sd[prefix + pre + "mlp.c_proj.weight"] = r(w)
            b = t(pre + "mlp.c_proj.bias")
            b[massive] += signs * np.float32(6.0 * i)
            sd[prefix + pre + "mlp.c_proj.bias"] = r(b)
        if i <= 4:
            for name, amp, const in (("attn.out_proj", 1.2, 0.5), ("mlp.c_proj", 0.5, 0.4)):
                w = t(pre + name + ".weight")
                v = rng.standard_normal(w.shape[1], dtype=np.float32) * np.float32(amp / np.sqrt(w.shape[1])) * (w.std() * np.sqrt(w.shape[1]))
                sd[prefix + pre + name + ".weight"] = r(w + ones[:, None] * v[None, :])
                sd[prefix + pre + name + ".bias"] = r(t(pre + name + ".bias") + np.float32(const))
        for ln in ("ln_1", "ln_2"):
            g = t(pre + ln + ".weight")
            g *= np.exp(0.35 * rng.standard_normal(width, dtype=np.float32))            # log-normal spread of the gains (trained: 0.3 - 3)
            g[massive] = rng.uniform(0.05, 0.5, size=3).astype(np.float32)
            t(pre + ln + ".bias")[:] += 0.3 * rng.standard_normal(width, dtype=np.float32)
        w = t(pre + "attn.in_proj_weight")
        w[: 2 * width] *= np.float32(qk_gain[i])
        sd[prefix + pre + "attn.in_proj_weight"] = r(w)
        b = t(pre + "attn.in_proj_bias")
        b[: 2 * width] *= np.float32(qk_gain[i])
        sd[prefix + pre + "attn.in_proj_bias"] = r(b)
    return massive


# q / k gains of stats="trained" per block, calibrated (tools/clip_stats.py) so that the scaled scores q.k / sqrt(dh) of the synthetic tiles have a standard
# deviation of 2 - 4 in the trunk and 6 - 10 in block 11, the regime of the released checkpoints (near one-hot rows in the last blocks)
TRAINED_QK_GAIN = {"visual": [3.0] * 11 + [4.7], "text": [2.0] * 12}


def make_clip_state_dict(arch: str = "ViT-B/32", seed: int = 0, sharpen: float = 2.0,
                         text_tower: bool = True, stats: str = "init") -> dict:
    """State dict with the OpenAI CLIP key names for a ViT-B model, values from `seed`.

    Standard deviations follow `CLIP.initialize_parameters` (`model_explainability.py:418-452`);
    biases / LayerNorm affine parameters are made non-trivial on purpose so parity tests exercise them.
    stats = "init": the benign statistics of a fresh initialisation (every golden before g29).  stats = "trained": the same draw edited towards the
    residual-stream statistics of a trained checkpoint (`_trained_tower`: massive-activation channels, per-row DC offsets, peaked softmax, class
    embedding norm >> patch norm) - the numbers the fp16 operand paths are hardest on (goldens g29).
    """
    assert stats in ("init", "trained")
    a = CLIP_ARCHS[arch]
    rng = np.random.default_rng(seed)
    n = lambda *s: rng.standard_normal(s, dtype=np.float32)
    W, p = a["width"], a["patch"]
    g = IMAGE_RES // p
    sd = {}
    scale = W ** -0.5
    sd["visual.conv1.weight"] = _f16(n(W, 3, p, p) * (3 * p * p) ** -0.5)
    sd["visual.class_embedding"] = scale * n(W)
    sd["visual.positional_embedding"] = scale * n(g * g + 1, W)
    sd["visual.ln_pre.weight"] = 1.0 + 0.1 * n(W)
    sd["visual.ln_pre.bias"] = 0.05 * n(W)
    proj_std = (W ** -0.5) * ((2 * a["layers"]) ** -0.5)
    for i in range(a["layers"]):
        _block(rng, f"visual.transformer.resblocks.{i}.", W, sd, W ** -0.5, proj_std, (2 * W) ** -0.5, sharpen)
    sd["visual.ln_post.weight"] = 1.0 + 0.1 * n(W)
    sd["visual.ln_post.bias"] = 0.05 * n(W)
    sd["visual.proj"] = _f16(scale * n(W, a["embed"]))
    if text_tower:
        TW = a["twidth"]
        sd["token_embedding.weight"] = 0.02 * n(VOCAB_SIZE, TW)
        sd["positional_embedding"] = 0.01 * n(CONTEXT_LENGTH, TW)
        tproj_std = (TW ** -0.5) * ((2 * a["tlayers"]) ** -0.5)
        for i in range(a["tlayers"]):
            _block(rng, f"transformer.resblocks.{i}.", TW, sd, TW ** -0.5, tproj_std, (2 * TW) ** -0.5, 2.0)
        sd["ln_final.weight"] = 1.0 + 0.1 * n(TW)
        sd["ln_final.bias"] = 0.05 * n(TW)
        sd["text_projection"] = _f16(n(TW, a["embed"]) * TW ** -0.5)
        sd["logit_scale"] = np.float32(np.log(1 / 0.07)) * np.ones((), dtype=np.float32)
    if stats == "trained":
        trng = np.random.default_rng(seed + 104729)              # its own stream: the "init" draw above is unchanged
        sd["visual.class_embedding"] = sd["visual.class_embedding"] * np.float32(12.0)
        _trained_tower(sd, "visual.", "visual.ln_pre", W, a["layers"], trng, TRAINED_QK_GAIN["visual"])
        if text_tower:
            _trained_tower(sd, "", None, a["twidth"], a["tlayers"], trng, TRAINED_QK_GAIN["text"])
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# SemAbs3D (point MLP + ResidualUNet3D + implicit decoder), key names as `net.py:319-381`
# ----------------------------------------------------------------------------------------------

def unet_layer_plan(in_channels: int, out_channels: int, f_maps: int, num_levels: int):
    """Channel plan of `ResidualUNet3D` (`unet3d.py:529-580`): list of (key_prefix, kind, cin, cout)."""
    fm = [f_maps * 2 ** k for k in range(num_levels)]
    plan = []
    for i, c in enumerate(fm):
        cin = in_channels if i == 0 else fm[i - 1]
        pre = f"encoders.{i}.basic_module."
        plan += [(pre + "conv1.", "gcr", cin, c), (pre + "conv2.", "gcr", c, c), (pre + "conv3.", "gc", c, c)]
    rf = fm[::-1]
    for i in range(len(rf) - 1):
        plan.append((f"decoders.{i}.upsampling.upsample.", "convT", rf[i], rf[i + 1]))
        pre = f"decoders.{i}.basic_module."
        c = rf[i + 1]
        plan += [(pre + "conv1.", "gcr", c, c), (pre + "conv2.", "gcr", c, c), (pre + "conv3.", "gc", c, c)]
    plan.append(("final_conv.", "conv1", fm[0], out_channels))
    return plan


def make_unet_state_dict(seed: int, in_channels: int, out_channels: int, f_maps: int, num_levels: int) -> dict:
    """Default initialisation of a stand-alone `ResidualUNet3D` (torch-default-like scales; GroupNorm affine = (1, 0) like nn.GroupNorm)."""
    rng = np.random.default_rng(seed)
    u = lambda fan_in, *s: ((rng.random(s, dtype=np.float32) * 2 - 1) / np.sqrt(fan_in)).astype(np.float32)
    sd = {}
    for pre, kind, cin, cout in unet_layer_plan(in_channels, out_channels, f_maps, num_levels):
        if kind in ("gcr", "gc"):
            sd[pre + "groupnorm.weight"] = np.ones(cin, np.float32)
            sd[pre + "groupnorm.bias"] = np.zeros(cin, np.float32)
            sd[pre + "conv.weight"] = u(cin * 27, cout, cin, 3, 3, 3)
        elif kind == "convT":
            sd[pre + "weight"] = u(cout * 27, cin, cout, 3, 3, 3)
            sd[pre + "bias"] = u(cout * 27, cout)
        else:
            sd[pre + "weight"] = u(cin, cout, cin, 1, 1, 1)
            sd[pre + "bias"] = u(cin, cout)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def make_semabs3d_state_dict(seed: int = 0, unet_num_channels: int = 16, unet_f_maps: int = 16,
                             unet_num_groups: int = 8, unet_num_levels: int = 6,
                             pts_feat_extractor_hidden_dim: int = 128, pts_feature_dim: int = 1,
                             output_dim: int = 1, decoder_concat_xyz_pts: bool = True, stats: str = "init") -> dict:
    """fp32 state dict for `SemAbs3D` (`net.py:319-381`) with torch-default-like init scales.
    stats = "trained" (golden g30): GroupNorm affine parameters with the spread of a trained checkpoint - gains log-normal around 1 (0.2 - 4, two channels per
    layer near zero, one at 6 x), offsets of +-0.5 - and convolution kernels with a few dominant output channels (x 4), so that the hi / lo operand split and the
    folded GroupNorm tables see uneven channel magnitudes; drawn from a separate stream, the "init" draw is unchanged."""
    assert stats in ("init", "trained")
    rng = np.random.default_rng(seed)
    n = lambda *s: rng.standard_normal(s, dtype=np.float32)
    u = lambda fan_in, *s: ((rng.random(s, dtype=np.float32) * 2 - 1) / np.sqrt(fan_in)).astype(np.float32)
    sd = {"steps": np.zeros(1, dtype=np.float32)}
    H = pts_feat_extractor_hidden_dim
    dims = [(pts_feature_dim + 3, H), (H, H), (H, unet_num_channels)]
    for li, (i, o) in zip((0, 2, 4), dims):
        sd[f"pts_feat_extractor.{li}.weight"] = u(i, o, i)
        sd[f"pts_feat_extractor.{li}.bias"] = u(i, o)
    for pre, kind, cin, cout in unet_layer_plan(unet_num_channels, unet_num_channels, unet_f_maps, unet_num_levels):
        k = "vol_feature_extractor." + pre
        if kind in ("gcr", "gc"):
            sd[k + "groupnorm.weight"] = 1.0 + 0.1 * n(cin)
            sd[k + "groupnorm.bias"] = 0.05 * n(cin)
            sd[k + "conv.weight"] = u(cin * 27, cout, cin, 3, 3, 3) * np.float32(1.7)
        elif kind == "convT":
            sd[k + "weight"] = u(cout * 27 / 8, cin, cout, 3, 3, 3)
            sd[k + "bias"] = u(cin * 27, cout)
        else:
            sd[k + "weight"] = u(cin, cout, cin, 1, 1, 1)
            sd[k + "bias"] = u(cin, cout)
    if stats == "trained":
        trng = np.random.default_rng(seed + 15485863)
        for pre, kind, cin, cout in unet_layer_plan(unet_num_channels, unet_num_channels, unet_f_maps, unet_num_levels):
            k = "vol_feature_extractor." + pre
            if kind in ("gcr", "gc"):
                g = np.exp(0.6 * trng.standard_normal(cin)).astype(np.float32)
                g[trng.choice(cin, 2, replace=False)] = np.float32(0.02)
                g[trng.choice(cin, 1)] = np.float32(6.0)
                sd[k + "groupnorm.weight"] = g
                sd[k + "groupnorm.bias"] = (0.5 * trng.standard_normal(cin)).astype(np.float32)
                w = sd[k + "conv.weight"]
                w[trng.choice(cout, max(1, cout // 16), replace=False)] *= np.float32(4.0)
    hid = unet_num_channels
    din = hid + 3 * int(decoder_concat_xyz_pts)
    sd["visual_sampler.mlp.0.weight"] = u(din, hid, din)
    sd["visual_sampler.mlp.0.bias"] = u(din, hid)
    sd["visual_sampler.mlp.2.weight"] = u(hid, output_dim, hid)
    sd["visual_sampler.mlp.2.bias"] = u(hid, output_dim)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


DEFAULT_LABELS = [
    "chair", "table", "lamp", "sofa", "bed", "mirror", "carpet", "wall",
    "floor", "door", "window", "shelf", "plant", "television", "cushion", "cabinet",
]
DEFAULT_PROMPT = "a photograph of a {} in a home."

VOOL_RELATIONS = ["in", "behind", "in front of", "on the left of", "on the right of", "on", "[pad]"]


def make_semabsvool_state_dict(seed: int = 0, pointing_dim: int = 64, unet_num_channels: int = 16, **semabs_kwargs) -> dict:
    """fp32 state dict for `SemAbsVOOL` (`net.py:469-504`): completion_net.* (a SemAbs3D), spatial_sampler.mlp.{0,2}.*,
    relation_embeddings.<name>, steps."""
    rng = np.random.default_rng(seed + 7919)
    u = lambda fan_in, *s: ((rng.random(s, dtype=np.float32) * 2 - 1) / np.sqrt(fan_in)).astype(np.float32)
    # SemAbsVOOL swallows decoder_concat_xyz_pts itself (net.py:470-483), so its completion_net is built WITHOUT the xyz concat
    sd = {"completion_net." + k: v for k, v in make_semabs3d_state_dict(seed=seed, unet_num_channels=unet_num_channels,
                                                                         decoder_concat_xyz_pts=False, **semabs_kwargs).items()}
    hid = 2 * unet_num_channels
    din = hid + 3
    extra = {"steps": np.zeros(1, dtype=np.float32),
             "spatial_sampler.mlp.0.weight": u(din, hid, din), "spatial_sampler.mlp.0.bias": u(din, hid),
             "spatial_sampler.mlp.2.weight": u(hid, pointing_dim, hid), "spatial_sampler.mlp.2.bias": u(hid, pointing_dim)}
    rel = {}
    for name in VOOL_RELATIONS:                              # values drawn in this order (the goldens depend on it) ...
        rel["relation_embeddings." + name] = rng.standard_normal(pointing_dim, dtype=np.float32)
    for k in sorted(rel):                                    # ... but listed in the reference's parameter order: nn.ParameterDict sorts its keys, and
        extra[k] = rel[k]                                    # `Lamb(net.parameters())` / optimizer state dicts are positional (utils.py:264-266, 289)
    sd.update({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in extra.items()})
    return sd
