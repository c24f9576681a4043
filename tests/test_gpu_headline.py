"""GPU: the BASELINE.json headline relevancy shape against the UNMODIFIED reference.

`tests/golden/g16_headline_{aug0,aug5}.npz` hold `ClipWrapper.get_clip_saliency` of the reference (CLIP/clip/__init__.py:103-236, CPU fp32,
run once in the build container by `tests/golden/gen_golden.py g16`) for one 480 x 480 synthetic image, ViT-B/16, 16 labels, the "ours"
configuration: `aug0` with augmentations=0 (408 tile forwards), `aug5` with the five augmented copies injected as fixed images
(`synth_jitter`, identical pixels on both sides) = 6 images, 2 448 forwards - the benchmarked workload.  Stored: the [::4, ::4] subsample of
the fp32 maps, 8 full rows, per-label max / sum, and the reference's zero-shot text weights (used here as `w_text`, so the BPE table is
not needed on the GPU box).

Both ViT batch sizes are exercised: chunk_tiles = 2448 (what bench.py times: ONE 482 256-row batch, ragged last wave of GEMM tiles, 11 GB
workspace) and 220; the maps must be bit-identical.  Tolerance: RELATIVE L-infinity = max|ours - ref| / max|ref| per run and per label (the reference's own
fp16 canvases quantise at 2^-11 = 4.9e-4 relative per add); the absolute error (north_star's bar is 1e-3 ABSOLUTE) is ~2e-6."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd.synth import synth_jitter, synth_rgb

pytestmark = pytest.mark.gpu

# The result is bit-reproducible (same kernels, same summation order for every batch size): bars = 1.3 x the MEASURED values, capped at the bars VERDICT r5
# item 2 set for the benchmarked path (1.0e-3 overall, 1.2e-3 per label).  Measured on MI355X, default precision (relative / worst label / absolute):
#   round 4: 6.0e-4 / 9.4e-4 / 2.9e-6 (aug0), 9.0e-4 / 1.06e-3 / 2.2e-6 (aug5);  round 5 (LayerNorm fold, un-centred): 7.52e-4 / 9.50e-4, 1.026e-3 / 1.148e-3;
#   round 6 (fold centred on the row's previous mean; last block + VJP chain on [hi | lo] operands): 7.35e-4 / 8.18e-4 / 3.5e-6, 6.84e-4 / 9.29e-4 / 1.7e-6.
# What is left is the fp16 rounding of the 11 trunk blocks' GEMM operands (tests/test_vit_precision_budget.py); an L-infinity over 3.7 M cells of ~2 448 averaged
# tile errors moves by +-15 % between equally accurate roundings.
MEASURED = {"aug0": (7.35e-4, 8.184e-4, 3.497e-6), "aug5": (6.84e-4, 9.291e-4, 1.695e-6)}
REL_LINF_BOUND = {t: min(1.3 * m[0], 1.0e-3) for t, m in MEASURED.items()}
PER_LABEL_BOUND = {t: min(1.3 * m[1], 1.2e-3) for t, m in MEASURED.items()}
ABS_LINF_BOUND = {t: 1.3 * m[2] for t, m in MEASURED.items()}
# precision = "parity" (q, k as fp16 hi + lo pairs in the scores; +10 ms per scene): measured 9.36e-4 / 1.096e-3 (aug0), 6.55e-4 / 8.71e-4 (aug5).  Since round 6 it is
# no longer systematically closer than the default path: q | k is one of six operand classes that carry the trunk's remaining error in about equal parts.
PARITY_MEASURED = {"aug0": (9.355e-4, 1.096e-3), "aug5": (6.554e-4, 8.708e-4)}
PARITY_REL_BOUND = {t: min(1.3 * m[0], 1.2e-3) for t, m in PARITY_MEASURED.items()}
PARITY_PER_LABEL_BOUND = {t: min(1.3 * m[1], 1.4e-3) for t, m in PARITY_MEASURED.items()}


@pytest.fixture(scope="module")
def wrapper():
    from semabs_amd.clip import ClipWrapper
    from semabs_amd.weights import make_clip_state_dict
    sd = make_clip_state_dict("ViT-B/16", 0, text_tower=False)

    def make(chunk, precision=None):
        ClipWrapper.engine = None
        ClipWrapper("ViT-B/16", state_dict=sd, chunk_tiles=chunk, max_labels=16, precision=precision)
        return ClipWrapper

    return make


def _run(W, g, tag):
    from semabs_amd.clip import saliency_configs
    img = synth_rgb(480, 480, seed=0)
    cfg = saliency_configs["ours"](480)
    if tag == "aug0":
        images = W.make_images(img, 0)
    else:
        images = W.make_images(img, cfg["augmentations"], jittered_images=[synth_jitter(img, k) for k in range(cfg["augmentations"])])
    w_text = torch.from_numpy(g["text"]).cuda().contiguous()
    maps = W.relevancy_device(images, w_text, cfg["cropping_augmentations"], cfg["horizontal_flipping"], cfg["positive_attn_only"])
    torch.cuda.synchronize()
    return maps.cpu().numpy()


@pytest.mark.parametrize("tag", ["aug0", "aug5"])
def test_headline_maps_vs_reference(golden, wrapper, tag):
    g = golden(f"g16_headline_{tag}")
    res = {}
    for chunk in (2448, 220):
        m = _run(wrapper(chunk), g, tag)
        assert m.shape == (16, 480, 480) and np.isfinite(m).all()
        ref_max = float(g["absmax"].max())
        e_sub = np.abs(m[:, ::4, ::4] - g["sub"]).max()
        e_rows = np.abs(m[:, g["rows_idx"], :] - g["rows"]).max()
        rel = float(max(e_sub, e_rows)) / ref_max
        per_label = (np.abs(m[:, ::4, ::4] - g["sub"]).reshape(16, -1).max(1) / g["absmax"]).max()
        sums = np.abs(m.astype(np.float64).reshape(16, -1).sum(1) - g["sums"]).max() / np.abs(g["sums"]).max()
        print(f"headline {tag} chunk {chunk}: abs L-inf {max(e_sub, e_rows):.3e}  relative L-inf {rel:.3e}  worst per-label relative {per_label:.3e}  "
              f"map-sum relative {sums:.3e}  (max|ref| {ref_max:.3e})")
        assert rel <= REL_LINF_BOUND[tag], rel
        assert per_label <= PER_LABEL_BOUND[tag], per_label
        assert max(e_sub, e_rows) <= ABS_LINF_BOUND[tag]                   # (BASELINE.json's bar is 1e-3 absolute)
        res[chunk] = m
    # the maps must not depend on the ViT batch size (same kernels, same per-row arithmetic; was tools/chunk_equiv.py)
    assert np.array_equal(res[2448], res[220]), float(np.abs(res[2448] - res[220]).max())


@pytest.mark.parametrize("tag", ["aug0", "aug5"])
def test_headline_maps_parity_precision(golden, wrapper, tag):
    """precision = "parity" (q and k as fp16 hi + lo pairs in the attention scores) on the headline shape, printed next to the default path."""
    g = golden(f"g16_headline_{tag}")
    m = _run(wrapper(2448, "parity"), g, tag)
    ref_max = float(g["absmax"].max())
    rel = float(max(np.abs(m[:, ::4, ::4] - g["sub"]).max(), np.abs(m[:, g["rows_idx"], :] - g["rows"]).max())) / ref_max
    per_label = (np.abs(m[:, ::4, ::4] - g["sub"]).reshape(16, -1).max(1) / g["absmax"]).max()
    print(f"headline {tag} precision=parity: relative L-inf {rel:.3e}  worst per-label relative {per_label:.3e}")
    assert rel <= PARITY_REL_BOUND[tag] and per_label <= PARITY_PER_LABEL_BOUND[tag], (rel, per_label)
