"""Relevancy storage format.  Golden g19 = the reference's `generate_saliency_helper` and `get_scene_patches` EXECUTED (compiled
from its source in the build container, containers stubbed; tests/golden/gen_golden.py g19): oracle vs golden on the CPU, HIP kernels vs golden
on the GPU."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import relevancy_io as orio


def _cases(g):
    for tag in "abc":
        L, H, W, h, w = (int(v) for v in g[f"{tag}_dims"])
        yield tag, L, H, W, h, w


def test_oracle_relevancy_io_matches_reference_expressions(golden):
    g = golden("g19_relevancy_storage")
    for tag, L, H, W, h, w in _cases(g):
        stored, tf = orio.pack_relevancy(torch.from_numpy(g[f"{tag}_maps"]), torch.from_numpy(g[f"{tag}_feats"]), (h, w))
        assert np.array_equal(stored.numpy(), g[f"{tag}_stored"])
        assert np.array_equal(tf.numpy(), g[f"{tag}_tf"])
        loaded = orio.unpack_relevancy(stored, (H, W), rows=g[f"{tag}_rows"].tolist(), mean_index=L, scale=50.0)
        assert np.array_equal(loaded.numpy(), g[f"{tag}_loaded50"])
        assert g[f"{tag}_names"].tolist() == [f"label{i}" for i in range(L)] + ["mean"]
        assert np.array_equal(g[f"{tag}_label_features"], g[f"{tag}_tf"][g[f"{tag}_rows"]])          # the loader's feature rows


@pytest.mark.gpu
def test_hip_relevancy_io_vs_oracle(golden):
    from semabs_amd.relevancy_io import pack_relevancy, unpack_relevancy
    g = golden("g19_relevancy_storage")
    for tag, L, H, W, h, w in _cases(g):
        maps, feats = torch.from_numpy(g[f"{tag}_maps"]), torch.from_numpy(g[f"{tag}_feats"])
        stored, tf = pack_relevancy(maps.cuda(), feats.cuda(), (h, w))
        ref_s, ref_f = g[f"{tag}_stored"], g[f"{tag}_tf"]
        assert np.array_equal(stored[:L].cpu().numpy(), ref_s[:L])                     # nearest-exact: a pure gather, bit-exact
        assert np.abs(stored[L].cpu().numpy() - ref_s[L]).max() <= 1e-6 * np.abs(ref_s[L]).max() + 1e-9      # mean row: fp32 sum order
        assert np.abs(tf.cpu().numpy() - ref_f).max() <= 2e-7
        rows = g[f"{tag}_rows"].tolist()
        loaded = unpack_relevancy(torch.from_numpy(ref_s).cuda(), (H, W), rows=rows, mean_index=L, scale=50.0)
        ref_l = g[f"{tag}_loaded50"]
        # bilinear: same indices and fp32 weights as ATen; its vectorised CPU kernel's rounding order is not reproduced to the bit
        # (measured: 36 % of the pixels differ, by at most 2.3e-6 of the maximum)
        assert np.abs(loaded.cpu().numpy() - ref_l).max() <= 5e-6 * np.abs(ref_l).max()
        # all rows, no mean subtraction, unit scale
        full = unpack_relevancy(torch.from_numpy(ref_s).cuda(), (H, W))
        ref_full = orio.unpack_relevancy(torch.from_numpy(ref_s), (H, W))
        assert np.abs(full.cpu().numpy() - ref_full.numpy()).max() <= 5e-6 * float(ref_full.abs().max())


@pytest.mark.gpu
def test_hip_relevancy_io_full_size_roundtrip():
    """480 x 480 x 16 labels stored at 128 x 128 and loaded back: rows commute with the mean subtraction (linearity of the bilinear map)."""
    from semabs_amd.relevancy_io import pack_relevancy, unpack_relevancy
    rng = np.random.default_rng(9)
    maps = torch.from_numpy((rng.standard_normal((16, 480, 480)) * 0.01).astype(np.float32)).cuda()
    feats = torch.from_numpy(rng.standard_normal((16, 512)).astype(np.float32)).cuda()
    stored, tf = pack_relevancy(maps, feats, (128, 128))
    assert tuple(stored.shape) == (17, 128, 128) and tuple(tf.shape) == (17, 512)
    assert float((tf.norm(dim=1) - 1).abs().max()) < 1e-6
    a = unpack_relevancy(stored, (480, 480), rows=range(16), mean_index=16, scale=50.0)
    b = unpack_relevancy(stored, (480, 480), rows=range(16), scale=50.0) - unpack_relevancy(stored, (480, 480), rows=[16], scale=50.0)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    assert float(a.mean(dim=0).abs().max()) <= 1e-5 * float(a.abs().max())           # the stored mean row is the mean of the stored rows
