"""GPU: the BASELINE.json headline relevancy shape against the UNMODIFIED reference.

`tests/golden/g16_headline_{aug0,aug5}.npz` hold `ClipWrapper.get_clip_saliency` of the reference (CLIP/clip/__init__.py:103-236, CPU fp32,
run once in the build container by `tests/golden/gen_golden.py g16`) for one 480 x 480 synthetic image, ViT-B/16, 16 labels, the "ours"
configuration: `aug0` with augmentations=0 (408 tile forwards), `aug5` with the five augmented copies injected as fixed images
(`synth_jitter`, identical pixels on both sides) = 6 images, 2 448 forwards - the benchmarked workload.  Stored: the [::4, ::4] subsample of
the fp32 maps, 8 full rows, per-label max / sum, and the reference's zero-shot text weights (used here as `w_text`, so the BPE table is
not needed on the GPU box).

Both ViT batch sizes are exercised: chunk_tiles = 2448 (what bench.py times: ONE 482 256-row batch, ragged last wave of GEMM tiles, 11 GB
workspace) and 220.  Tolerance: RELATIVE L-infinity = max|ours - ref| / max|ref| per run; measured on MI355X 6.0e-4 (aug0) and 9.0e-4 (aug5) (fp16 MFMA operands
against the reference's fp32 CPU arithmetic; the reference's own fp16 canvases quantise at 2^-11 = 4.9e-4 relative per add), i.e. 2.9e-6 /
2.2e-6 absolute; asserted at 1.3 x the measured values (the result is bit-reproducible).  Worst single label relative to its own maximum: 9.4e-4 / 1.06e-3."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd.synth import synth_jitter, synth_rgb

pytestmark = pytest.mark.gpu

# The result is bit-reproducible (same kernels, same summation order for every batch size), so the bars sit at 1.3 x the MEASURED values (VERDICT r3
# item 3; they were 3 x): relative L-infinity 6.0e-4 (aug0) / 9.0e-4 (aug5), worst label relative to its own maximum 9.4e-4 / 1.06e-3, absolute 2.9e-6 /
# 2.2e-6.  BASELINE bar: 1e-3 ABSOLUTE on maps whose max is 4.8e-3.  Where the 9.0e-4 comes from: tests/test_vit_precision_budget.py (by block and by
# operand class: the fp16 rounding of q and k, 1.5e-3 of the 2.05e-3 per tile, then the LayerNorm-1 output, 9e-4).
REL_LINF_BOUND = {"aug0": 7.8e-4, "aug5": 1.17e-3}
PER_LABEL_BOUND = {"aug0": 1.23e-3, "aug5": 1.38e-3}
ABS_LINF_BOUND = {"aug0": 3.8e-6, "aug5": 2.9e-6}


@pytest.fixture(scope="module")
def wrapper():
    from semabs_amd.clip import ClipWrapper
    from semabs_amd.weights import make_clip_state_dict
    sd = make_clip_state_dict("ViT-B/16", 0, text_tower=False)

    def make(chunk):
        ClipWrapper.engine = None
        ClipWrapper("ViT-B/16", state_dict=sd, chunk_tiles=chunk, max_labels=16)
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
