"""GPU: the relevancy path on weights with TRAINED-checkpoint statistics against the UNMODIFIED reference (goldens g29).

Every other relevancy golden runs seeded random-init weights, whose residual stream is benign (no outlier channels, zero row means, near-uniform softmax).
`make_clip_state_dict(stats="trained")` edits the same draw towards what the released CLIP checkpoints show (tools/clip_stats.py prints it): three
massive-activation channels at 60 - 85 x the spread of the others, a per-row DC offset of ~4 sigma of the bulk, scaled attention scores with sigma 3 - 8
in the trunk and 8 - 9 in block 11 (mean row maximum of the softmax 0.6 - 0.8), class embedding 12 x the patch embedding - the numbers the fp16 operand
paths (fp16 q / k, the fp16 copy of x * gamma feeding the folded LayerNorm GEMMs) are hardest on.  The goldens are the reference's own autograd /
`get_clip_saliency` outputs on those weights, text tower included (tests/golden/gen_golden.py g29).

What these cases found (round 6) and what was changed for them - all measured on MI355X, relative L-infinity = max|ours - ref| / max|ref|:
  * per tile, round-5 code: 0.7 - 8.5e-3 (random-init weights: 0.75 - 1.8e-3); maps 1.0 - 2.5e-3; headline shape 1.68e-3 (random init: 1.03e-3).
  * stage-by-stage against the oracle (tools/stage_errors.py): every intermediate within 5e-4 EXCEPT the kept softmax row of block 11, 3 - 4e-3 absolute: with CLS
    scores up to ~60 the fp16 rounding of the last LayerNorm output alone moves it by 2e-3 (d p = p (1 - p) d s, d s ~ |s| 2^-11).  Fix: every fp16 A operand
    behind the trunk (last block + VJP chain, ~1 % of the flops) is an [hi | lo] pair now (clip/vit.py head_split; +1.1 ms per scene).
  * the un-centred fp16(x * gamma) copy of the LayerNorm fold loses bits to the row's DC offset (GEMM level: 6.4e-4 vs 4.1e-4 at 4 sigma, 3.2e-3 at 20 sigma):
    the producer centres on the row's previous mean now (clip/vit.py ln_center; free).
  * now: per tile 0.55 - 3.0e-3 (what is left is the trunk's fp16-operand noise, ~5e-4 of the bulk spread of x, amplified by the same peaked softmax: the
    budget in tests/test_vit_precision_budget.py), maps 0.77 - 1.28e-3, headline shape 5.7e-4; random-init per tile 0.75 - 1.1e-3, headline 6.8e-4 (aug5).
Bars: conftest.bar = 1.3 x the error each case measured, under a ceiling of 4.0e-3 per tile (3 tiles: an L-infinity of a few hundred values) and 1.5 - 2.0e-3 for maps."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import preprocess as op
from semabs_amd.synth import synth_rgb
from semabs_amd.weights import DEFAULT_LABELS, DEFAULT_PROMPT, make_clip_state_dict

pytestmark = pytest.mark.gpu

PER_TILE_CEILING = 4.0e-3


def _engine(arch, chunk, precision=None, max_labels=4):
    from semabs_amd.clip import ClipWrapper
    sd = make_clip_state_dict(arch, 0, stats="trained")
    ClipWrapper.engine = None
    ClipWrapper(arch, state_dict=sd, chunk_tiles=chunk, max_labels=max_labels, precision=precision)
    return ClipWrapper


def _tiles(n, seed):
    sizes = [120, 80, 60, 30, 97]
    return torch.from_numpy(np.stack([op.preprocess_tile(synth_rgb(sizes[i % 5], sizes[i % 5], seed=seed + i)) for i in range(n)]))


@pytest.mark.parametrize("arch,tag,reps,precision", [("ViT-B/32", "b32", 1, None), ("ViT-B/32", "b32", 16, None), ("ViT-B/32", "b32", 16, "parity"),
                                                     ("ViT-B/16", "b16", 1, None), ("ViT-B/16", "b16", 4, None), ("ViT-B/16", "b16", 4, "parity")])
def test_vit_gradcam_trained_statistics(golden, bar, arch, tag, reps, precision):
    """3 tiles x 4 labels, positive_attn_only True / False, vs the reference's autograd result.  reps = 1: 150 / 591 token rows - the small-batch launch
    sequence (LayerNorm kernels, ring GEMM kernel); reps = 16 / 4: the same tiles repeated to >= 2 048 rows, i.e. the benchmarked sequence (persistent GEMMs
    with the LayerNorm folded in) - every copy of a tile must reproduce the golden."""
    CW = _engine(arch, 3 * reps, precision)
    g = golden(f"g29_vit_{tag}")
    tiles = _tiles(3, 7).repeat(reps, 1, 1, 1).cuda()
    w_text = torch.from_numpy(g["w_text"]).T.contiguous().cuda()
    for pos in (True, False):
        rel, logits, feat = CW.engine.gradcam_tiles(tiles, w_text, pos)
        ref = np.tile(g[f"rel_pos{int(pos)}"], (1, reps, 1, 1))
        top = np.abs(ref).max()
        err = np.abs(rel.cpu().numpy() - ref).max() / top
        print(f"{arch} trained statistics, {3 * reps} tiles, precision {precision}, pos={pos}: relative L-inf {err:.3e} (max|ref| {top:.3e})")
        assert bar(f"rel_pos{int(pos)}", err, PER_TILE_CEILING)
    np.testing.assert_allclose(feat.cpu().numpy(), np.tile(g["feat"], (reps, 1)), rtol=0, atol=5e-3 * np.abs(g["feat"]).max())
    np.testing.assert_allclose(logits.cpu().numpy(), np.tile(g["logits"], (reps, 1)), rtol=0, atol=5e-3 * np.abs(g["logits"]).max() + 0.05)
    probs = CW.engine._workspace()["probs"][:3].cpu().numpy()
    np.testing.assert_allclose(probs, g["probs_cls"], rtol=3e-2, atol=2e-5)


def test_text_tower_trained_statistics(golden):
    """Zero-shot weights of the text tower (its own massive channels / DC offsets) from the reference's token ids vs the reference's weights."""
    CW = _engine("ViT-B/32", 8)
    g = golden("g29_vit_b32")
    w = CW.text.zeroshot_weights(torch.from_numpy(g["tokens"]), 4, 1).cpu().numpy()
    ref = g["w_text"].T
    err = np.abs(w - ref).max() / np.abs(ref).max()
    print(f"text tower, trained statistics: relative L-inf {err:.3e}")
    assert err <= 5e-3


@pytest.mark.parametrize("precision", [None, "parity"])
@pytest.mark.parametrize("arch,name,H,pos", [("ViT-B/32", "b32_ours120", 120, True), ("ViT-B/32", "b32_ours120_signed", 120, False),
                                             ("ViT-B/16", "b16_ours240", 240, True)])
def test_end_to_end_maps_trained_statistics(golden, bar, arch, name, H, pos, precision):
    """uint8 image -> fp32 maps ("ours", augmentations = 0: 408 tile forwards, 4 labels), the whole HIP path in ONE batch (the benchmarked launch sequence)."""
    from semabs_amd.clip import saliency_configs
    CW = _engine(arch, 408, precision)
    g = golden("g29_e2e")
    cfg = dict(saliency_configs["ours"](H), augmentations=0, positive_attn_only=pos)
    w_text = torch.from_numpy(g[f"{name}_text"]).contiguous().cuda()
    img = torch.from_numpy(synth_rgb(H, H, seed=42)).cuda()[None].contiguous()
    maps = CW.relevancy_device(img, w_text, cfg["cropping_augmentations"], cfg["horizontal_flipping"], cfg["positive_attn_only"])
    ref = g[f"{name}_maps"]
    top = np.abs(ref).max()
    err = np.abs(maps.cpu().numpy() - ref).max() / top
    per_label = (np.abs(maps.cpu().numpy() - ref).reshape(4, -1).max(1) / np.abs(ref).reshape(4, -1).max(1)).max()
    print(f"{arch}/{name} trained statistics, precision {precision}: relative L-inf {err:.3e}, worst label {per_label:.3e} (max|ref| {top:.3e})")
    assert bar("maps_rel", err, 2.0e-3)


@pytest.mark.parametrize("precision", [None, "parity"])
def test_headline_shape_trained_statistics(golden, bar, precision):
    """480 x 480, ViT-B/16, 16 labels, "ours" with augmentations = 0 (408 forwards) on the trained-statistics weights vs the reference's `get_clip_saliency`."""
    from semabs_amd.clip import saliency_configs
    CW = _engine("ViT-B/16", 408, precision, max_labels=16)
    g = golden("g29_headline_aug0")
    cfg = dict(saliency_configs["ours"](480), augmentations=0)
    w_text = torch.from_numpy(g["text"]).contiguous().cuda()
    img = torch.from_numpy(synth_rgb(480, 480, seed=0)).cuda()[None].contiguous()
    maps = CW.relevancy_device(img, w_text, cfg["cropping_augmentations"], cfg["horizontal_flipping"], cfg["positive_attn_only"]).cpu().numpy()
    sub, rows = g["sub"], g["rows"]
    top = float(g["absmax"].max())
    d = max(np.abs(maps[:, ::4, ::4] - sub).max(), np.abs(maps[:, g["rows_idx"], :] - rows).max())
    per_label = max((np.abs(maps[l, ::4, ::4] - sub[l]).max() / g["absmax"][l]) for l in range(16))
    print(f"headline shape, trained statistics, precision {precision}: relative L-inf {d / top:.3e}, worst label {per_label:.3e} (max|ref| {top:.3e}, absolute {d:.2e})")
    assert bar("maps_rel", d / top, 1.5e-3)
    assert d <= 1e-3                                          # north_star's absolute bar
