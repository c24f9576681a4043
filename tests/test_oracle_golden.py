"""CPU tests: the oracle (our restatement) against golden vectors captured from the reference import
(tests/golden/gen_golden.py).  This is what pins the oracle; the HIP path is then tested against the oracle."""
import os

import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from conftest import sha
from oracle import geometry as og
from oracle import preprocess as op
from oracle import relevancy as orl
from oracle import semabs3d as os3
from semabs_amd.synth import synth_rgb, synth_scene
from semabs_amd.weights import DEFAULT_PROMPT, make_clip_state_dict, make_semabs3d_state_dict

SCENE_BOUNDS = [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]]


# ---- a1/a2 -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,H,W,cfgname,dim,n_img", [
    ("480", 480, 480, "ours", 480, 2), ("120", 120, 120, "ours", 120, 2), ("256x192", 256, 192, "ours", 256, 2),
    ("96c", 96, 96, "chefer_et_al", 96, 1), ("100x130", 100, 130, "ours", 100, 2)])
def test_tiling(golden, tag, H, W, cfgname, dim, n_img):
    g = golden("g1_tiling")
    cfg = orl.saliency_configs[cfgname](dim)
    table = orl.tile_table(H, W, n_img, cfg["cropping_augmentations"])
    assert np.array_equal(table, g[f"table_{tag}"])
    assert np.array_equal(table[:, 3], g[f"sizes_{tag}"])
    counts = orl.tile_counts(H, W, table)
    # the reference creates a count canvas for every crop_aug, even one that produced no tile
    keys = [k for k in g[f"count_keys_{tag}"] if k in counts]
    assert keys == list(counts.keys())
    for k, c in counts.items():
        assert c.astype(np.float64).sum() == g[f"count_{tag}_{k}_sum"]
        assert np.array_equal(c[::7, ::5], g[f"count_{tag}_{k}_sub"])


def test_ours_480_tile_count():
    cfg = orl.saliency_configs["ours"](480)
    assert len(orl.tile_table(480, 480, 6, cfg["cropping_augmentations"])) == 1224


# ---- a3 ------------------------------------------------------------------------------------------
def test_resize_matches_pillow():
    from PIL import Image
    rng = np.random.default_rng(1)
    for ts in (480, 320, 240, 120, 224, 80, 30, 7):
        img = rng.integers(0, 256, (ts, ts, 3), dtype=np.uint8)
        ref = np.array(Image.fromarray(img).resize((224, 224), Image.BICUBIC))
        assert np.array_equal(op.resize_bicubic_u8(img), ref), ts


@pytest.mark.parametrize("ts", [480, 320, 240, 120, 80, 30, 224])
def test_preprocess(golden, ts):
    g = golden("g2_preprocess")
    t = op.preprocess_tile(synth_rgb(ts, ts, seed=1000 + ts))
    assert np.array_equal(sha(t), g[f"ts{ts}_sha"])          # bit-exact fp32


# ---- a4-a8 -----------------------------------------------------------------------------------------
def _tiles(n, seed):
    sizes = [120, 80, 60, 30, 97]
    return torch.from_numpy(np.stack([op.preprocess_tile(synth_rgb(sizes[i % 5], sizes[i % 5], seed=seed + i))
                                      for i in range(n)]))


@pytest.mark.parametrize("arch,tag", [("ViT-B/32", "b32"), ("ViT-B/16", "b16")])
def test_vit_gradcam(golden, arch, tag):
    g = golden(f"g3g4_vit_{tag}")
    sd = make_clip_state_dict(arch, 0, text_tower=False)
    tiles = _tiles(3, 7)
    assert abs(tiles.double().sum().item() - g["tiles_sum"]) < 1e-6
    w_text = torch.from_numpy(g["w_text"])
    with torch.no_grad():
        if "pos_emb" in g:
            pe = orl.interpolate_positional_emb(sd["visual.positional_embedding"], g["pos_emb"].shape[0])
            assert np.array_equal(pe.numpy(), g["pos_emb"])
        feat, last = orl.vit_forward(sd, tiles)
        np.testing.assert_allclose(feat.numpy(), g["feat"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(last["probs"][:, :, 0, :].numpy(), g["probs_cls"], rtol=1e-4, atol=1e-7)
        for pos in (True, False):
            rel, logits = orl.gradcam_tiles(sd, tiles, w_text, pos)
            ref = g[f"rel_pos{int(pos)}"]
            scale = np.abs(ref).max()
            assert np.abs(rel.numpy() - ref).max() <= 2e-5 * scale + 1e-9, (np.abs(rel.numpy() - ref).max(), scale)
        np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=1e-4)
    assert g["grad_l1_noncls_absmax"] == 0.0      # only the CLS query row carries gradient


@pytest.mark.parametrize("arch,tag", [("ViT-B/32", "b32"), ("ViT-B/16", "b16")])
def test_vit_gradcam_trained_statistics(golden, arch, tag):
    """g29: the same closed form against the reference's AUTOGRAD result on weights with trained-checkpoint statistics (massive-activation channels,
    per-row DC offsets, peaked softmax - `make_clip_state_dict(stats="trained")`), zero-shot weights from the reference's tokenizer + text tower."""
    g = golden(f"g29_vit_{tag}")
    sd = make_clip_state_dict(arch, 0, stats="trained")
    tiles = _tiles(3, 7)
    assert abs(tiles.double().sum().item() - g["tiles_sum"]) < 1e-6
    with torch.no_grad():
        w_text = orl.zeroshot_weights(sd, torch.from_numpy(g["tokens"]), 4, 1)
        np.testing.assert_allclose(w_text.numpy(), g["w_text"], rtol=2e-4, atol=2e-6)
        w_text = torch.from_numpy(g["w_text"])
        feat, last = orl.vit_forward(sd, tiles)
        np.testing.assert_allclose(feat.numpy(), g["feat"], rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose(last["probs"][:, :, 0, :].numpy(), g["probs_cls"], rtol=2e-4, atol=1e-7)
        assert float(last["probs"][:, :, 0, :].max(-1).values.mean()) > 0.3          # the CLS rows ARE peaked (random-init weights: ~0.05)
        for pos in (True, False):
            rel, logits = orl.gradcam_tiles(sd, tiles, w_text, pos)
            ref = g[f"rel_pos{int(pos)}"]
            scale = np.abs(ref).max()
            assert np.abs(rel.numpy() - ref).max() <= 5e-5 * scale + 1e-9, (np.abs(rel.numpy() - ref).max(), scale)
        np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=2e-4, atol=2e-4)


def test_text_weights(golden):
    g = golden("g7_text")
    sd = make_clip_state_dict("ViT-B/32", 0)
    with torch.no_grad():
        for tag, nt in (("t1", 1), ("t3", 3)):
            w = orl.zeroshot_weights(sd, torch.from_numpy(g[f"{tag}_tokens"]), 4, nt)
            np.testing.assert_allclose(w.numpy(), g[f"{tag}_weights"], rtol=1e-4, atol=1e-6)


# ---- a9 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,cfgname", [("ours120", "ours"), ("chefer96", "chefer_et_al"), ("ours56_g14", "ours"), ("ours64x48", "ours")])
def test_aggregate(golden, tag, cfgname):
    g = golden("g5_aggregate")
    H, gg, L, aug, flip, W = (int(v) for v in g[f"{tag}_meta"])
    cfg = orl.saliency_configs[cfgname](int(H))
    table = orl.tile_table(H, W, aug + 1, cfg["cropping_augmentations"])
    rel = torch.from_numpy(g[f"{tag}_rel"])
    if flip:
        rel = (rel + torch.flip(torch.from_numpy(g[f"{tag}_rel_flip"]), dims=[-1])) / 2
    out = orl.aggregate(rel, table, H, W, tile_sizes=[a["tile_size"] for a in cfg["cropping_augmentations"]])
    assert np.array_equal(out.numpy(), g[f"{tag}_maps"])       # same ops in the same order: bit-exact


# ---- a10 end to end -------------------------------------------------------------------------------
@pytest.mark.parametrize("arch,tag,name,H", [("ViT-B/32", "b32", "chefer96", 96), ("ViT-B/32", "b32", "ours96", 96),
                                             ("ViT-B/16", "b16", "two_scale64", 64)])
def test_end_to_end(golden, arch, tag, name, H):
    g = golden(f"g6_e2e_{tag}")
    sd = make_clip_state_dict(arch, 0, text_tower=False)
    if name == "ours96":
        cfg = dict(orl.saliency_configs["ours"](96), augmentations=0)
    elif name == "chefer96":
        cfg = orl.saliency_configs["chefer_et_al"](96)
    else:
        cfg = dict(orl.saliency_configs["chefer_et_al"](64), horizontal_flipping=True,
                   cropping_augmentations=[{"tile_size": 64, "stride": 16}, {"tile_size": 32, "stride": 8}])
    w_text = torch.from_numpy(g[f"{name}_text"]).T.contiguous()
    with torch.no_grad():
        maps = orl.relevancy_maps(sd, [synth_rgb(H, H, seed=42)], w_text, **cfg)
    ref = g[f"{name}_maps"]
    # canvases are fp16: a ~1e-7 relative difference upstream (closed form vs autograd) can flip one fp16
    # rounding, i.e. 2^-10 relative on that element; everything else agrees to fp32 noise
    err = np.abs(maps.numpy() - ref)
    assert (err <= 1.0e-3 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all(), (err.max(), np.abs(ref).max())
    assert (err > 2e-5 * np.abs(ref).max()).mean() < 1e-3


# ---- a11-a13 ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,hw,S", [("48", 48, 32), ("480", 480, 128)])
def test_geometry(golden, tag, hw, S):
    g = golden("g8_geometry")
    sc = synth_scene(hw, hw, seed=5)
    pts = og.get_pointcloud(sc["depth"], sc["cam_intr"], sc["cam_pose"])
    pts32 = pts.astype(np.float32)
    assert np.array_equal(sha(pts32), g[f"{tag}_pts32_sha"])
    mask = og.filter_pts_bounds(pts32, np.array(SCENE_BOUNDS))
    assert mask.sum() == g[f"{tag}_mask_count"] and np.array_equal(sha(mask), g[f"{tag}_mask_sha"])
    assert mask.mean() > 0.5
    flat = og.flatten_idxs(og.points_grid_idxs(pts32, SCENE_BOUNDS, (S, S, S)), (S, S, S))
    assert np.array_equal(sha(flat.astype(np.int64)), g[f"{tag}_flat_sha"])
    fr = og.check_pts_in_frustum(pts32[::3].astype(np.float64) * 1.01, sc["depth"].shape, sc["cam_pose"], sc["cam_intr"])
    assert fr.sum() == g[f"{tag}_frustum_count"] and np.array_equal(sha(fr), g[f"{tag}_frustum_sha"])


# ---- a19 -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,hw,S", [("16", 48, 16), ("32", 64, 32), ("128", 480, 128)])
def test_tsdf(golden, tag, hw, S):
    g = golden("g11_tsdf")
    vs = (SCENE_BOUNDS[1][0] - SCENE_BOUNDS[0][0]) / S
    tv = og.TSDFVolume(np.array(SCENE_BOUNDS).T, vs)
    assert np.array_equal(tv._vol_dim, g[f"{tag}_dim"])
    for k in range(2 if S <= 32 else 1):
        sc = synth_scene(hw, hw, seed=6 if k == 0 else 16)
        tv.integrate(sc["rgb"], sc["depth"], sc["cam_intr"], sc["cam_pose"])
    assert np.array_equal(sha(tv._last["pix"]), g[f"{tag}_pix_sha"])
    assert np.array_equal(sha(tv._tsdf_vol_cpu), g[f"{tag}_tsdf_sha"])
    assert np.array_equal(sha(tv._weight_vol_cpu), g[f"{tag}_weight_sha"])
    assert np.array_equal(sha(tv._color_vol_cpu), g[f"{tag}_color_sha"])
    assert (tv._weight_vol_cpu > 0).sum() == g[f"{tag}_n_obs"] > 0


# ---- a14-a18 ---------------------------------------------------------------------------------------
def semabs_inputs(S, N, M, P, seed):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = (lo + (hi - lo) * rng.random((1, N, 3))).astype(np.float32)
    xyz[0, : N // 8] = xyz[0, N // 8: 2 * (N // 8)] + np.float32(1e-3)
    feat = (rng.standard_normal((1, P, N, 1)) * 0.5).astype(np.float32)
    q = (lo - 0.05 + (hi - lo + 0.1) * rng.random((1, P, M, 3))).astype(np.float32)
    return xyz, feat, q


@pytest.mark.parametrize("stats,name", [("init", "g9_semabs3d"), ("trained", "g30_semabs3d_trained")])
def test_semabs3d(golden, stats, name):
    g = golden(name)
    S, N, M, P, seed, wseed = g["meta"]
    sd = make_semabs3d_state_dict(seed=int(wseed), stats=stats)
    xyz, feat, q = semabs_inputs(S, N, M, P, int(seed))
    taps = {}
    with torch.no_grad():
        out = os3.semabs3d_forward(sd, torch.from_numpy(xyz), torch.from_numpy(feat), torch.from_numpy(q),
                                   SCENE_BOUNDS, (S, S, S), taps=taps)
    np.testing.assert_allclose(taps["scatter"].numpy()[:, :, ::3, ::3, ::3], g["scatter_sub"], rtol=1e-5, atol=1e-6)
    assert (taps["scatter"][:, 0] != 0).sum().item() == g["scatter_nonzero"]
    for k in [k for k in taps if k.startswith(("enc", "dec"))]:
        v = taps[k]
        np.testing.assert_allclose(v.numpy()[:, ::max(1, v.shape[1] // 8), ::2, ::2, ::2], g[f"tap_{k}_sub"],
                                   rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(taps["unet"].numpy()[:, :, ::3, ::3, ::3], g["unet_sub"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=1e-4)


@pytest.mark.skipif(os.environ.get("SEMABS_SKIP_SLOW") == "1", reason="slow")
def test_unet128(golden):
    g = golden("g10_unet128")
    sd = make_semabs3d_state_dict(seed=int(g["meta"][1]))
    rng = np.random.default_rng(int(g["meta"][0]))
    x = np.zeros((1, 16, 128, 128, 128), np.float32)
    occ = rng.random((128, 128, 128)) < 0.03
    x[0][:, occ] = rng.standard_normal((16, int(occ.sum()))).astype(np.float32)
    with torch.no_grad():
        y = os3.unet_forward(sd, torch.from_numpy(x), 6).numpy()
    np.testing.assert_allclose(y.reshape(-1)[g["si"]], g["y_s"], rtol=1e-3, atol=1e-4)
    assert abs(np.abs(y.astype(np.float64)).sum() - g["y_abs"]) <= 1e-5 * g["y_abs"]


def test_oracle_semabs3d_tsdf_input_vs_reference(golden):
    """network_inputs = ["saliency", "tsdf"]: oracle vs the reference module's own output (g17, its torch-seeded parameters stored alongside)."""
    import torch
    from oracle import semabs3d as os3
    g = golden("g17_semabs3d_tsdf")
    S = int(g["meta"][0])
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    with torch.no_grad():
        out = os3.semabs3d_forward(sd, torch.from_numpy(g["xyz"]), torch.from_numpy(g["feat"]), torch.from_numpy(g["q"]),
                                   [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]], (S, S, S), num_levels=int(g["meta"][4]), tsdf_vol=torch.from_numpy(g["tsdf"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-4, atol=2e-5)


def test_oracle_ovssc_post_mask_vs_executed_reference(golden):
    """f1 glue: the reference's process_batch_ovssc + get_sample_points executed from source (g18) vs oracle.scene.{sample_points, ovssc_post_mask}."""
    import hashlib
    import torch
    from oracle import scene as osc
    from semabs_amd.synth import SCENE_BOUNDS, synth_ovssc_logits, synth_scene
    g = golden("g18_process_batch_ovssc")
    S, C, hw, seed, _ = (int(v) for v in g["meta"])
    pts = osc.sample_points((S, S, S), SCENE_BOUNDS)
    assert np.array_equal(np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8), g["points_sha"])
    assert np.array_equal(pts[::997], g["points_sub"])
    vols = osc.ovssc_post_mask(synth_ovssc_logits(torch.from_numpy(pts), C), synth_scene(hw, hw, seed=seed), SCENE_BOUNDS, (S, S, S))
    ref = np.unpackbits(g["packed"], axis=1)[:, : S ** 3].reshape(C, S, S, S).astype(np.float32)
    assert np.array_equal(vols.reshape(C, -1).sum(1).astype(np.int64), g["counts"])
    assert np.array_equal(vols, ref)


def test_g26_process_batch_vool_fixture_is_the_stand_in_on_the_reference_lattice(golden):
    """f5 glue: the reference's process_batch_vool executed from source (g26).  Without a GPU: the stored volumes are exactly the closed-form
    stand-in evaluated on the oracle's sampling lattice with the row / relation of each description - i.e. the fixture pins what the test says."""
    from oracle import scene as osc
    from semabs_amd.synth import SCENE_BOUNDS, synth_vool_logits
    g = golden("g26_process_batch_vool")
    S, D, n_in, chunk = (int(v) for v in g["meta"])
    pts = osc.sample_points((S, S, S), SCENE_BOUNDS)
    assert np.array_equal(sha(pts), g["points_sha"])
    want = np.stack([synth_vool_logits(torch.from_numpy(pts), float(g["tgt"][d]), float(g["ref"][d]), str(g["relations"][d])).numpy().reshape(S, S, S) for d in range(D)])
    assert np.array_equal(want, g["volumes"])
    assert [int(c) for c in g["chunks"]] == [chunk] * (S ** 3 // chunk) + ([S ** 3 % chunk] if S ** 3 % chunk else [])


# ---- a2: colour jitter of the augmentation copies (g28 = torchvision's PIL path executed with Pillow on fixed orders / factors) --------------
def test_color_jitter_oracle_vs_pillow_golden(golden):
    g = golden("g28_color_jitter")
    img = synth_rgb(int(g["meta"][0]), int(g["meta"][1]), seed=int(g["meta"][2]))
    for o, f, want, sub in zip(g["orders"], g["factors"], g["sha"], g["sub"]):
        out = op.color_jitter(img, o, f)
        assert np.array_equal(out[::5, ::5], sub), (o, f)
        assert np.array_equal(sha(out), want), (o, f)
    for opid in range(4):
        for k, f in enumerate(g["single_f"][opid]):
            assert np.array_equal(sha(op.JITTER_OPS[opid](img, float(f))), g["single_sha"][opid][k]), (opid, f)
    # both HSV conversions on a 2^21-colour lattice (the generator checked all 2^24 against the oracle when it wrote the fixture)
    c = np.arange(1 << 24, dtype=np.uint32)
    lat = np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], axis=-1).astype(np.uint8).reshape(256, 256, 256, 3)[::2, ::2, ::2].reshape(-1, 1, 3)
    assert np.array_equal(sha(op.rgb_to_hsv_u8(lat)), g["lattice_hsv_sha"])
    assert np.array_equal(sha(op.hsv_to_rgb_u8(lat)), g["lattice_rgb_sha"])
    assert op.hue_shift_u8(-0.05) == 244 and op.hue_shift_u8(0.1) == 25 and op.hue_shift_u8(0.0) == 0
