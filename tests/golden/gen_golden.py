"""Generate the golden vectors under tests/golden/*.npz by running the UNMODIFIED reference
(`/root/reference`, imported through the stand-ins in refimport.py) on seeded inputs.

Run here (the build container), once:   python tests/golden/gen_golden.py [group ...]
The GPU box never runs this (it has no /root/reference); it only reads the committed .npz files.
Fixtures hold inputs' seeds/recipes + the reference's outputs (data only, no reference source).
Weights come from `semabs_amd.weights` (seeded numpy PCG64), loaded into the reference modules.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refimport  # noqa: E402
import semabs_amd  # noqa: E402,F401
from semabs_amd.weights import DEFAULT_PROMPT, make_semabs3d_state_dict  # noqa: E402
from semabs_amd.synth import synth_rgb, synth_scene  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(os.cpu_count())


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1e3:.1f} kB")


def digest(a: np.ndarray) -> np.ndarray:
    """sha256 of the raw bytes as uint8[32] (for bit-exact integer outputs too big to commit)."""
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def sample_idx(n, k, seed=123):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False))


# ----------------------------------------------------------------------------------------------------
def g1_tiling(rc):
    """create_tiles geometry: tile table, tile_sizes, counts for 480^2 / 120^2 / 256x192 (ours + chefer)."""
    W = rc.ClipWrapper
    out = {}
    for tag, (H, Wd), cfgname, dim in [("480", (480, 480), "ours", 480), ("120", (120, 120), "ours", 120),
                                        ("256x192", (256, 192), "ours", 256), ("96c", (96, 96), "chefer_et_al", 96),
                                        ("100x130", (100, 130), "ours", 100)]:
        cfg = rc.saliency_configs[cfgname](dim)
        img = np.zeros((H, Wd, 3), np.uint8)
        # preprocessing 1 224 tiles is slow and irrelevant here: stub the per-tile preprocess
        real_pre = W.clip_gradcam.preprocess
        W.clip_gradcam.preprocess = lambda pil: torch.zeros(1)
        aug = 1 if cfgname == "ours" else 0
        tiles, _, counts, sizes = W.create_tiles(img=img, augmentations=aug,
                                                 cropping_augmentations=cfg["cropping_augmentations"])
        W.clip_gradcam.preprocess = real_pre
        # tiles are (slice(None), slice(x, x+ts), slice(y, y+ts)); image index is implied by order
        per_img = len(tiles) // (aug + 1)
        table = np.array([(i // per_img, t[1].start, t[2].start, t[1].stop - t[1].start)
                          for i, t in enumerate(tiles)], dtype=np.int32).reshape(-1, 4)
        out[f"table_{tag}"] = table
        out[f"sizes_{tag}"] = np.asarray(sizes, np.int32)
        out[f"count_keys_{tag}"] = np.asarray(list(counts.keys()), np.int32)
        for k, c in counts.items():
            c = c.numpy()
            out[f"count_{tag}_{k}_sum"] = np.float64(c.astype(np.float64).sum())
            out[f"count_{tag}_{k}_sub"] = c[::7, ::5].copy()
    save("g1_tiling", **out)


def g2_preprocess(rc):
    """clip_gradcam.preprocess (Resize bicubic 224 -> ToTensor -> Normalize) on synthetic square crops."""
    from PIL import Image
    W = rc.ClipWrapper
    out = {}
    for ts in (480, 320, 240, 120, 80, 30, 224):
        crop = synth_rgb(ts, ts, seed=1000 + ts)
        t = W.clip_gradcam.preprocess(Image.fromarray(crop)).numpy()
        out[f"ts{ts}_sum"] = np.float64(t.astype(np.float64).sum())
        out[f"ts{ts}_sub"] = t[:, ::9, ::7].copy()
        out[f"ts{ts}_sha"] = digest(t)
    save("g2_preprocess", **out)


def _tiles_from_seed(rc, n, seed):
    from PIL import Image
    W = rc.ClipWrapper
    sizes = [120, 80, 60, 30, 97]
    return torch.stack([W.clip_gradcam.preprocess(Image.fromarray(synth_rgb(sizes[i % 5], sizes[i % 5], seed=seed + i)))
                        for i in range(n)])


def g3_g4_vit(arch, tag):
    """ViT forward + ClipGradcam.interpret (autograd) on 3 tiles x 3 labels, positive_attn_only True/False."""
    rc = refimport.load_reference_clip(arch, seed=0)
    W = rc.ClipWrapper
    gc = W.clip_gradcam
    labels = ["chair", "table", "lamp"]
    gc.templates = [DEFAULT_PROMPT]
    gc.set_classes(labels)
    w_text = torch.cat([gc.class_to_language_feature[c] for c in labels], dim=1)
    tiles = _tiles_from_seed(rc, 3, seed=7)
    out = {"w_text": w_text.numpy(), "tiles_sum": np.float64(tiles.double().sum().item())}
    with torch.no_grad():
        vis = gc.model.visual
        x = vis.conv1(tiles)
        T = x.shape[-1] * x.shape[-2] + 1
        if T != 50:
            from CLIP.clip.auxiliary import interpolate_positional_emb
            out["pos_emb"] = interpolate_positional_emb(vis.positional_embedding, T).numpy()
        out["feat"] = gc.model.encode_image(tiles).numpy()
    for pos in (True, False):
        gc.positive_attn_only = pos
        rel = gc(x=tiles, o=labels)
        out[f"rel_pos{int(pos)}"] = rel.detach().numpy()
    blk = list(gc.model.visual.transformer.resblocks.children())[-1]
    probs = blk.attn_probs.detach()
    out["probs_cls"] = probs.view(3, 12, T, T)[:, :, 0, :].numpy()
    # logits + raw gradient of the CLS row for label 1
    feats = gc.model.encode_image(tiles)
    feats = feats / feats.norm(dim=-1, keepdim=True)
    logits = 100.0 * feats @ w_text
    out["logits"] = logits.detach().numpy()
    blk = list(gc.model.visual.transformer.resblocks.children())[-1]
    g = torch.autograd.grad(logits.sum(dim=0)[1], [blk.attn_probs])[0]
    out["grad_l1_cls"] = g.view(3, 12, T, T)[:, :, 0, :].numpy()
    out["grad_l1_noncls_absmax"] = np.float64(g.view(3, 12, T, T)[:, :, 1:, :].abs().max().item())
    save(f"g3g4_vit_{tag}", **out)
    return rc


def g5_aggregate(rc):
    """get_clip_saliency_convolve with clip_gradcam replaced by a seeded fake: pins flip-average,
    bilinear upsampling, fp16 canvases, count normalisation and the mean over scales."""
    W = rc.ClipWrapper
    real = W.clip_gradcam
    out = {}
    for tag, H, cfgname, g, L, aug in [("ours120", 120, "ours", 7, 3, 1), ("chefer96", 96, "chefer_et_al", 7, 2, 0),
                                        ("ours56_g14", 56, "ours", 14, 2, 0), ("ours64x48", 64, "ours", 7, 2, 0)]:
        Wd = 48 if tag == "ours64x48" else H          # non-square: the 64-pixel scale has no tile, yet stays in the mean
        cfg = rc.saliency_configs[cfgname](H)

        class Fake:
            positive_attn_only = False
            preprocess = staticmethod(lambda pil: torch.zeros(3, 4, 4))

            def __init__(self):
                self.rng = np.random.default_rng(99)
                self.calls = []

            def __call__(self, x, o):
                r = torch.from_numpy((self.rng.standard_normal((len(o), len(x), g, g)) * 0.01).astype(np.float32))
                self.calls.append(r)
                return r

        fake = Fake()
        W.clip_gradcam = fake
        labels = [f"l{i}" for i in range(L)]
        maps = W.get_clip_saliency_convolve(img=np.zeros((H, Wd, 3), np.uint8), text_labels=labels,
                                            horizontal_flipping=cfg["horizontal_flipping"],
                                            positive_attn_only=True, augmentations=aug,
                                            cropping_augmentations=cfg["cropping_augmentations"])
        n_pass = 2 if cfg["horizontal_flipping"] else 1
        per = len(fake.calls) // n_pass
        out[f"{tag}_rel"] = torch.cat(fake.calls[:per], dim=1).numpy()
        if n_pass == 2:
            out[f"{tag}_rel_flip"] = torch.cat(fake.calls[per:], dim=1).numpy()
        out[f"{tag}_maps"] = maps.numpy()
        out[f"{tag}_meta"] = np.asarray([H, g, L, aug, int(cfg["horizontal_flipping"]), Wd], np.int32)
    W.clip_gradcam = real
    save("g5_aggregate", **out)


def g6_end_to_end(rc, arch, tag):
    """ClipWrapper.get_clip_saliency on a synthetic image (identity ColorJitter stub)."""
    W = rc.ClipWrapper
    out = {}
    labels = ["chair", "table", "lamp"]
    if arch == "ViT-B/32":
        runs = [("ours96", 96, dict(rc.saliency_configs["ours"](96), augmentations=0)),
                ("chefer96", 96, rc.saliency_configs["chefer_et_al"](96))]
    else:
        runs = [("two_scale64", 64, dict(rc.saliency_configs["chefer_et_al"](64), horizontal_flipping=True,
                                         cropping_augmentations=[{"tile_size": 64, "stride": 16},
                                                                 {"tile_size": 32, "stride": 8}]))]
    for name, H, cfg in runs:
        img = synth_rgb(H, H, seed=42)
        t = time.time()
        maps, feats = W.get_clip_saliency(img=img, text_labels=labels, prompts=[DEFAULT_PROMPT], **cfg)
        print(f"    {tag}/{name}: {time.time() - t:.1f}s  max|map| {maps.abs().max():.4g}")
        out[f"{name}_maps"] = maps.numpy()
        out[f"{name}_text"] = feats.numpy()
    save(f"g6_e2e_{tag}", **out)


def g7_text(rc):
    import CLIP.clip.clip_explainability as rexp
    W = rc.ClipWrapper
    gc = W.clip_gradcam
    out = {}
    labels = ["chair", "table", "pink make up bag", "brown modern upholstered chair in faux leather with wooden legs"]
    for tag, templates in [("t1", [DEFAULT_PROMPT]), ("t3", ["a photo of a {}.", "a bad photo of the {}.", DEFAULT_PROMPT])]:
        texts = [t.format(c) for c in labels for t in templates]
        out[f"{tag}_tokens"] = rexp.tokenize(texts).numpy()
        gc.templates = templates
        gc.set_classes(labels)
        out[f"{tag}_weights"] = torch.cat([gc.class_to_language_feature[c] for c in labels], dim=1).numpy()
    out["misc_tokens"] = rexp.tokenize(["Hello, World! it's 42 degrees", "a  b\tc", "don't you're we've"]).numpy()
    save("g7_text", **out)


# ----------------------------------------------------------------------------------------------------
SCENE_BOUNDS = [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]]


def g8_geometry():
    fusion, pc = refimport.load_reference_geometry()
    net, _ = refimport.load_reference_net()
    out = {}
    for tag, hw, S in [("48", 48, 32), ("480", 480, 128)]:
        sc = synth_scene(hw, hw, seed=5)
        pts = pc.get_pointcloud(sc["depth"], None, sc["cam_intr"], sc["cam_pose"])[0]
        pts32 = pts.astype(np.float32)
        mask = pc.filter_pts_bounds(pts32, np.array(SCENE_BOUNDS))
        vg = net.VirtualGrid(scene_bounds=np.array(SCENE_BOUNDS), grid_shape=(S, S, S), batch_size=1)
        idx = vg.get_points_grid_idxs(torch.from_numpy(pts32)[None])
        flat = vg.flatten_idxs(idx)[0].numpy()
        fr = pc.check_pts_in_frustum(pts32[::3].astype(np.float64) * 1.01, sc["depth"], sc["cam_pose"], sc["cam_intr"])
        out[f"{tag}_mask_sha"] = digest(mask)
        out[f"{tag}_mask_count"] = np.int64(mask.sum())
        out[f"{tag}_flat_sha"] = digest(flat.astype(np.int64))
        out[f"{tag}_pts32_sha"] = digest(pts32)
        out[f"{tag}_frustum_sha"] = digest(fr)
        out[f"{tag}_frustum_count"] = np.int64(fr.sum())
        if hw == 48:
            out["48_pts64"] = pts
            out["48_flat"] = flat
            out["48_mask"] = mask
        else:
            si = sample_idx(len(flat), 2048)
            out["480_si"] = si
            out["480_flat_s"] = flat[si]
            out["480_pts64_s"] = pts[si]
    save("g8_geometry", **out)


def g11_tsdf():
    fusion, pc = refimport.load_reference_geometry()
    out = {}
    for tag, hw, S in [("16", 48, 16), ("32", 64, 32), ("128", 480, 128)]:
        sc = synth_scene(hw, hw, seed=6)
        vs = (SCENE_BOUNDS[1][0] - SCENE_BOUNDS[0][0]) / S
        tv = fusion.TSDFVolume(vol_bnds=np.array(SCENE_BOUNDS).T.copy(), voxel_size=vs)
        tv._voxel_size = np.float64(tv._voxel_size)        # numba would type the python float as float64
        pix_rec = {}
        real_c2p = tv.cam2pix

        def rec(cam_pts, intr):
            p = real_c2p(cam_pts, intr)
            pix_rec["pix"] = p
            pix_rec["z"] = cam_pts[:, 2].copy()
            return p

        tv.cam2pix = rec
        t = time.time()
        n_int = 2 if S <= 32 else 1
        for k in range(n_int):
            sck = sc if k == 0 else synth_scene(hw, hw, seed=16)
            tv.integrate(sck["rgb"], sck["depth"], sck["cam_intr"], sck["cam_pose"], obs_weight=np.float64(1.0))
        print(f"    tsdf {tag}: {time.time() - t:.1f}s")
        tsdf, wgt, col = tv._tsdf_vol_cpu, tv._weight_vol_cpu, tv._color_vol_cpu
        out[f"{tag}_dim"] = np.asarray(tv._vol_dim, np.int32)
        out[f"{tag}_pix_sha"] = digest(pix_rec["pix"].astype(np.int64))     # of the LAST integrate
        out[f"{tag}_tsdf_sha"] = digest(tsdf)
        out[f"{tag}_weight_sha"] = digest(wgt)
        out[f"{tag}_color_sha"] = digest(col)
        out[f"{tag}_n_obs"] = np.int64((wgt > 0).sum())
        if S <= 32:
            out[f"{tag}_pix"] = pix_rec["pix"].astype(np.int64)
            out[f"{tag}_tsdf"] = tsdf
            out[f"{tag}_weight"] = wgt
            out[f"{tag}_color"] = col
        else:
            si = sample_idx(tsdf.size, 4096)
            out[f"{tag}_si"] = si
            out[f"{tag}_pix_s"] = pix_rec["pix"].astype(np.int64)[si]
            out[f"{tag}_tsdf_s"] = tsdf.reshape(-1)[si]
    save("g11_tsdf", **out)


def _semabs_inputs(S, N, M, P, seed):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = (lo + (hi - lo) * rng.random((1, N, 3))).astype(np.float32)
    xyz[0, : N // 8] = xyz[0, N // 8: 2 * (N // 8)] + np.float32(1e-3)  # make sure many voxels get >1 point
    feat = (rng.standard_normal((1, P, N, 1)) * 0.5).astype(np.float32)
    q = (lo - 0.05 + (hi - lo + 0.1) * rng.random((1, P, M, 3))).astype(np.float32)  # some outside bounds
    return xyz, feat, q


def g9_semabs3d(stats="init", name="g9_semabs3d"):
    """stats = "trained" -> g30_semabs3d_trained: the same forward on weights with trained-like GroupNorm gains / offsets and dominant channels."""
    net, unet3d = refimport.load_reference_net()
    S, N, M, P = 32, 3000, 2048, 2
    m = net.SemAbs3D(voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16,
                     unet_num_groups=8, unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True,
                     pts_feat_extractor_hidden_dim=128, reduce_method="max", output_dim=1, device="cpu",
                     decoder_concat_xyz_pts=True, batch_size=1)
    sd = make_semabs3d_state_dict(seed=3, stats=stats)
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    assert m.vg.reduce_method == "mean"
    xyz, feat, q = _semabs_inputs(S, N, M, P, seed=11)
    taps = {}
    hooks = []
    ue = m.vol_feature_extractor
    for i, e in enumerate(ue.encoders):
        hooks.append(e.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"enc{i}", o.detach().clone())))
    for i, d in enumerate(ue.decoders):
        hooks.append(d.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"dec{i}", o.detach().clone())))
    hooks.append(ue.register_forward_pre_hook(lambda mod, a: taps.__setitem__("scatter", a[0].detach().clone())))
    with torch.no_grad():
        out = m(input_xyz_pts=torch.from_numpy(xyz), input_feature_pts=torch.from_numpy(feat), tsdf_vol=None,
                output_xyz_pts=torch.from_numpy(q))
    res = {"out": out.numpy(), "unet_sub": m.visual_volumetric_features.numpy()[:, :, ::3, ::3, ::3].copy(),
           "scatter_sum": np.float64(taps["scatter"].double().sum().item()),
           "scatter_nonzero": np.int64((taps["scatter"][:, 0] != 0).sum().item()),
           "scatter_sub": taps["scatter"].numpy()[:, :, ::3, ::3, ::3].copy(),
           "meta": np.asarray([S, N, M, P, 11, 3], np.int32)}
    for k, v in taps.items():
        if k != "scatter":
            res[f"tap_{k}_sum"] = np.float64(v.double().sum().item())
            res[f"tap_{k}_abs"] = np.float64(v.double().abs().sum().item())
            res[f"tap_{k}_sub"] = v.numpy()[:, ::max(1, v.shape[1] // 8), ::2, ::2, ::2].copy()
    save(name, **res)


def g10_unet128():
    _, unet3d = refimport.load_reference_net()
    u = unet3d.ResidualUNet3D(in_channels=16, out_channels=16, f_maps=16, num_groups=8, num_levels=6)
    sd = make_semabs3d_state_dict(seed=3)
    pre = "vol_feature_extractor."
    u.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    u.eval()
    rng = np.random.default_rng(21)
    x = np.zeros((1, 16, 128, 128, 128), np.float32)
    occ = rng.random((128, 128, 128)) < 0.03                # sparse like a scattered point cloud
    x[0][:, occ] = rng.standard_normal((16, int(occ.sum()))).astype(np.float32)
    t = time.time()
    with torch.no_grad():
        y = u(torch.from_numpy(x)).numpy()
    print(f"    unet128: {time.time() - t:.1f}s")
    si = sample_idx(y.size, 8192)
    save("g10_unet128", y_s=y.reshape(-1)[si], si=si, y_sum=np.float64(y.astype(np.float64).sum()),
         y_abs=np.float64(np.abs(y.astype(np.float64)).sum()), meta=np.asarray([21, 3], np.int32))


GROUPS = ["g1", "g2", "g3b32", "g3b16", "g5", "g6b32", "g6b16", "g7", "g8", "g9", "g10", "g11"]

if __name__ == "__main__":
    want = sys.argv[1:] or GROUPS
    rc = None

    def clip32():
        global rc
        if rc is None or rc.ClipWrapper.clip_gradcam.clip_model_name != "ViT-B/32":
            rc = refimport.load_reference_clip("ViT-B/32", seed=0)
        return rc

    for gname in want:
        t0 = time.time()
        print(f"[{gname}]")
        if gname == "g1":
            g1_tiling(clip32())
        elif gname == "g2":
            g2_preprocess(clip32())
        elif gname == "g3b32":
            rc = g3_g4_vit("ViT-B/32", "b32")
        elif gname == "g3b16":
            rc = g3_g4_vit("ViT-B/16", "b16")
        elif gname == "g5":
            g5_aggregate(clip32())
        elif gname == "g6b32":
            g6_end_to_end(clip32(), "ViT-B/32", "b32")
        elif gname == "g6b16":
            rc = refimport.load_reference_clip("ViT-B/16", seed=0)
            g6_end_to_end(rc, "ViT-B/16", "b16")
        elif gname == "g7":
            g7_text(clip32())
        elif gname == "g8":
            g8_geometry()
        elif gname == "g9":
            g9_semabs3d()
        elif gname == "g10":
            g10_unet128()
        elif gname == "g11":
            g11_tsdf()
        print(f"  {time.time() - t0:.1f}s")


# ---- appended: VOOL forward + LAMB goldens (G12, forward / optimizer halves) --------------------------------------
def g12_vool_lamb():
    from semabs_amd.weights import make_semabsvool_state_dict
    net, _ = refimport.load_reference_net()
    S, N, M, D = 32, 3000, 1500, 3
    m = net.SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cpu", decoder_concat_xyz_pts=True,
                       voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8,
                       unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128,
                       reduce_method="max", batch_size=1)
    sd = make_semabsvool_state_dict(seed=3)
    m.load_state_dict(sd, strict=True)
    m.eval()
    xyz, feat, q = _semabs_inputs(S, N, M, 2 * D, seed=13)
    rel_names = [["behind"], ["on"], ["in front of"]]
    with torch.no_grad():
        out = m(output_xyz_pts=torch.from_numpy(q[:, :D]), spatial_relation_name=rel_names, input_xyz_pts=torch.from_numpy(xyz),
                input_target_saliency_pts=torch.from_numpy(feat[:, :D]), input_reference_saliency_pts=torch.from_numpy(feat[:, D:]),
                tsdf_vol=None)
    res = {"vool_out": out.numpy(), "vool_meta": np.asarray([S, N, M, D, 13, 3], np.int32)}
    # LAMB: 3 steps on a few tensors with seeded gradients (weight decay on, one all-zero tensor, one tiny tensor)
    sys.path.insert(0, refimport.REF)
    from arm.optim.lamb import Lamb
    rng = np.random.default_rng(77)
    shapes = [(64, 33), (7,), (128, 128), (5, 3, 3, 3, 3), (1,)]
    params = [torch.nn.Parameter(torch.from_numpy((rng.standard_normal(s) * (0.0 if i == 1 else 0.3)).astype(np.float32))) for i, s in enumerate(shapes)]
    opt = Lamb(params, lr=1e-3, weight_decay=1e-5)
    for step in range(3):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy((rng.standard_normal(p.shape) * 0.1).astype(np.float32))
        opt.step()
    for i, p in enumerate(params):
        res[f"lamb_w{i}"] = p.detach().numpy()
        st = opt.state[p]
        res[f"lamb_m{i}"] = st["exp_avg"].numpy()
        res[f"lamb_v{i}"] = st["exp_avg_sq"].numpy()
        res[f"lamb_stats{i}"] = np.asarray([float(st["weight_norm"]), float(st["adam_norm"]), float(st["trust_ratio"])], np.float32)
    save("g12_vool_lamb", **res)


if __name__ == "__main__" and "g12" in sys.argv[1:]:
    g12_vool_lamb()


# ---- appended: one VOOL training step on the reference (G13: loss, gradients, clipped LAMB update) ------------------
def g13_vool_train(S=32, N=3000, M=1500, D=3, name="g13_vool_train"):
    """One VOOL optimisation step of the unmodified reference (SemAbsVOOL + BCE-with-logits + clip_grad_norm_ + arm.optim.lamb.Lamb).
    g13: 32^3 (UNet level 5 = 1^3 voxels: GroupNorm singular there, gradients ill-conditioned); g20: 64^3, where that does not apply."""
    from semabs_amd.weights import make_semabsvool_state_dict
    net, _ = refimport.load_reference_net()
    sys.path.insert(0, refimport.REF)
    from arm.optim.lamb import Lamb
    m = net.SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cpu", decoder_concat_xyz_pts=True,
                       voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8,
                       unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128,
                       reduce_method="max", batch_size=1)
    m.load_state_dict(make_semabsvool_state_dict(seed=3), strict=True)
    m.train()
    xyz, feat, q = _semabs_inputs(S, N, M, 2 * D, seed=13)
    rel_names = [["behind"], ["on"], ["behind"]]                 # a repeated relation: its embedding gets two contributions
    rng = np.random.default_rng(131)
    label = (rng.random((1, D, M)) < 0.3).astype(np.float32)
    params = [(k, p) for k, p in m.named_parameters()]
    opt = Lamb([p for _, p in params], lr=1e-3, weight_decay=1e-5)
    before = {k: p.detach().clone() for k, p in params}
    out = m(output_xyz_pts=torch.from_numpy(q[:, :D]), spatial_relation_name=rel_names, input_xyz_pts=torch.from_numpy(xyz),
            input_target_saliency_pts=torch.from_numpy(feat[:, :D]), input_reference_saliency_pts=torch.from_numpy(feat[:, D:]),
            tsdf_vol=None)
    lab = torch.from_numpy(label)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, lab, weight=torch.ones_like(lab))
    opt.zero_grad()
    loss.backward()
    res = {"meta": np.asarray([S, N, M, D, 13, 3, 131], np.int32), "label": label.astype(np.uint8), "loss": np.float64(loss.item()),
           "logits": out.detach().numpy()}
    names, gnorm, has_grad = [], [], []
    for k, p in params:
        names.append(k)
        has_grad.append(p.grad is not None)
        gnorm.append(0.0 if p.grad is None else float(p.grad.double().norm()))
        if p.grad is not None:
            g = p.grad.detach().numpy().copy()           # a copy: clip_grad_norm_ below scales p.grad in place
            if g.size <= 4096:
                res["grad/" + k] = g
            else:
                si = sample_idx(g.size, 2048)
                res["gradidx/" + k] = si
                res["grads/" + k] = g.reshape(-1)[si]
    res["names"] = np.asarray(names)
    res["grad_norm"] = np.asarray(gnorm, np.float64)
    res["has_grad"] = np.asarray(has_grad)
    total = torch.nn.utils.clip_grad_norm_([p for _, p in params], 2.0)
    res["total_norm"] = np.float64(float(total))
    opt.step()
    dnorm = []
    for k, p in params:
        d = (p.detach() - before[k])
        dnorm.append(float(d.double().norm()))
        if p.numel() <= 4096:
            res["new/" + k] = p.detach().numpy()
        else:
            si = sample_idx(p.numel(), 2048)
            res["news/" + k] = p.detach().numpy().reshape(-1)[si]
    res["delta_norm"] = np.asarray(dnorm, np.float64)
    # a second configuration of the loss: the balanced BCE weights of utils.get_bce_weight (utils.py:727-749)
    # utils.py drags in tensorboardX / transformers / the dataset module; run just the one function, unmodified, from its source
    import ast
    tree = ast.parse(open(os.path.join(refimport.REF, "utils.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_bce_weight"][0]
    fn.decorator_list, fn.returns = [], None
    for a in fn.args.args:
        a.annotation = None
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "utils.py:get_bce_weight", "exec"), ns)
    w = ns["get_bce_weight"](output_label_pts=lab, balance_positive_negative=True)
    res["bce_weight_balanced_sum"] = np.float64(w.double().sum().item())
    res["bce_weight_balanced_sub"] = w.numpy()[:, :, ::50].copy()
    res["loss_balanced"] = np.float64(torch.nn.functional.binary_cross_entropy_with_logits(out.detach(), lab, weight=w).item())
    save(name, **res)


if __name__ == "__main__" and "g13" in sys.argv[1:]:
    g13_vool_train()
if __name__ == "__main__" and "g20" in sys.argv[1:]:
    g13_vool_train(S=64, N=12000, M=4000, D=3, name="g20_vool_train64")


# ---- appended: relevancy storage format (G14): the reference's own expressions at generate_relevancy.py:95-118 and dataset.py:821-871 -----
# ---- appended: evaluation metrics (G15): the reference's voxelize_points / prediction_analysis / iou run on seeded inputs ------------------
def g15_metrics():
    """utils.py cannot be imported here (tensorboardX / transformers / dataset), so the three functions are compiled from its source,
    unmodified except for the type annotations / decorators, with the reference's own VirtualGrid (torch_scatter stubbed)."""
    import ast
    net, _ = refimport.load_reference_net()
    tree = ast.parse(open(os.path.join(refimport.REF, "utils.py")).read())
    class _NumpyWithNAN:                                      # API drift: numpy 2 dropped the np.NAN alias the reference uses (utils.py:364,369)
        NAN = float("nan")

        def __getattr__(self, k):
            return getattr(np, k)

    ns = {"torch": torch, "np": _NumpyWithNAN(), "VirtualGrid": net.VirtualGrid}
    for name in ("iou", "prediction_analysis", "voxelize_points"):
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
        fn.decorator_list, fn.returns = [], None
        for a in fn.args.args:
            a.annotation = None
        exec(compile(ast.Module(body=[fn], type_ignores=[]), f"utils.py:{name}", "exec"), ns)
    rng = np.random.default_rng(150)
    B, P, N, S = 2, 3, 6000, 32
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = (lo - 0.05 + (hi - lo + 0.1) * rng.random((B, P, N, 3))).astype(np.float32)
    logits = rng.standard_normal((B, P, N)).astype(np.float32)
    label = (rng.random((B, P, N)) < 0.3)
    ignore = (rng.random((B, P, N)) < 0.1)
    ignore[1, 2] = True                                       # one fully ignored pair: the NaN branches
    pred = logits > 0.5
    pred[0, 1] = False                                        # no positive prediction: precision NaN
    vox = ns["voxelize_points"](prediction=torch.from_numpy(pred), label=torch.from_numpy(label), xyz_pts=torch.from_numpy(xyz), voxel_shape=(S, S, S),
                                scene_bounds=torch.tensor(SCENE_BOUNDS), ignore_pts=torch.from_numpy(ignore), device="cpu")
    pts = ns["prediction_analysis"](prediction=torch.from_numpy(pred), label=torch.from_numpy(label), ignore=torch.from_numpy(ignore))
    vst = ns["prediction_analysis"](**vox)
    res = {"meta": np.asarray([B, P, N, S]), "xyz": xyz, "pred": pred, "label": label, "ignore": ignore,
           "vox_prediction": vox["prediction"].numpy(), "vox_label": vox["label"].numpy(), "vox_ignore": vox["ignore"].numpy(),
           "iou_rows": ns["iou"](torch.from_numpy(pred), torch.from_numpy(label)).numpy()}
    for k, v in pts.items():
        res["point_" + k] = np.asarray(v, np.float64)
    for k, v in vst.items():
        res["voxel_" + k] = np.asarray(v, np.float64)
    save("g15_metrics", **res)


if __name__ == "__main__" and "g15" in sys.argv[1:]:
    g15_metrics()


# ---- appended (round 2): the headline relevancy shape, run through the unmodified reference ---------------------------
def g16_headline(which=("aug0", "aug5")):
    """ClipWrapper.get_clip_saliency at the BASELINE shape: 480 x 480, ViT-B/16, 16 labels, "ours" (CLIP/clip/__init__.py:103-236).
    aug0: augmentations=0 (1 image, 408 forwards).  aug5: the 5 augmented copies are injected (`synth_jitter`, the same pixels on
    both sides) through `ClipWrapper.jittering_transforms` -> 6 images, 2 448 forwards, the benchmarked workload.
    Stored per run: the [::4, ::4] subsample of the fp32 maps, 8 full rows, per-label max / sum / sha256."""
    from semabs_amd.weights import DEFAULT_LABELS
    from semabs_amd.synth import synth_jitter
    rc = refimport.load_reference_clip("ViT-B/16", seed=0)
    W = rc.ClipWrapper
    labels = list(DEFAULT_LABELS[:16])
    img = synth_rgb(480, 480, seed=0)
    for tag in which:
        cfg = dict(rc.saliency_configs["ours"](480))
        if tag == "aug0":
            cfg["augmentations"] = 0
        else:
            from PIL import Image
            queue = [Image.fromarray(synth_jitter(img, k)) for k in range(cfg["augmentations"])]
            it = iter(queue)
            W.jittering_transforms = lambda pil: next(it)
        t = time.time()
        maps, feats = W.get_clip_saliency(img=img, text_labels=labels, prompts=[DEFAULT_PROMPT], **cfg)
        dt = time.time() - t
        m = maps.numpy()
        print(f"    headline/{tag}: {dt:.1f}s on {os.cpu_count()} cores  max|map| {np.abs(m).max():.5g}", flush=True)
        rows = np.asarray([0, 61, 122, 183, 244, 305, 366, 479])
        save(f"g16_headline_{tag}", sub=m[:, ::4, ::4].copy(), rows_idx=rows, rows=m[:, rows, :].copy(),
             absmax=np.abs(m).reshape(16, -1).max(1), sums=m.astype(np.float64).reshape(16, -1).sum(1),
             sha=np.stack([digest(m[l]) for l in range(16)]), text=feats.numpy(), labels=np.asarray(labels),
             seconds=np.float64(dt), cores=np.int64(os.cpu_count()))


if __name__ == "__main__" and any(a.startswith("g16") for a in sys.argv[1:]):
    g16_headline(tuple(a.split(":")[1] for a in sys.argv[1:] if a.startswith("g16:")) or ("aug0", "aug5"))


# ---- appended (round 2): SemAbs3D with the TSDF network input, through the unmodified reference ------------------------------
def g17_semabs3d_tsdf():
    """net.SemAbs3D(network_inputs=["saliency", "tsdf"]).forward (net.py:346-357, 411-419) at 16^3, 3 UNet levels; the module's own
    (torch-seeded) parameters are stored next to the output."""
    net, _ = refimport.load_reference_net()
    S, N, M, P = 16, 2000, 300, 2
    torch.manual_seed(5)
    m = net.SemAbs3D(voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8,
                     unet_num_levels=3, network_inputs=["saliency", "tsdf"], use_pts_feat_extractor=True,
                     pts_feat_extractor_hidden_dim=128, reduce_method="max", output_dim=1, device="cpu", decoder_concat_xyz_pts=True,
                     batch_size=1).eval()
    rng = np.random.default_rng(1)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    xyz = (lo + (hi - lo) * rng.random((1, N, 3))).astype(np.float32)
    feat = (rng.standard_normal((1, P, N, 1)) * 0.5).astype(np.float32)
    q = (lo + (hi - lo) * rng.random((1, P, M, 3))).astype(np.float32)
    tsdf = rng.uniform(-1, 1, (1, S, S, S)).astype(np.float32)
    with torch.no_grad():
        out = m(input_xyz_pts=torch.from_numpy(xyz), input_feature_pts=torch.from_numpy(feat), tsdf_vol=torch.from_numpy(tsdf),
                output_xyz_pts=torch.from_numpy(q))
    res = {"out": out.numpy(), "xyz": xyz, "feat": feat, "q": q, "tsdf": tsdf, "meta": np.asarray([S, N, M, P, 3], np.int32)}
    for k, v in m.state_dict().items():
        res["sd::" + k] = v.numpy()
    save("g17_semabs3d_tsdf", **res)


if __name__ == "__main__" and "g17" in sys.argv[1:]:
    g17_semabs3d_tsdf()


# ---- appended (round 2): f1 / f2 glue pinned by EXECUTING the reference's functions (compiled from its source here, outputs only) -------
def _ref_functions(relpath, names, ns, in_class=None):
    """Compile the named FunctionDefs of a reference file (decorators and annotations stripped) into namespace `ns`."""
    import ast
    tree = ast.parse(open(os.path.join(refimport.REF, relpath)).read())
    body = tree.body
    if in_class is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == in_class][0].body
    for name in names:
        fn = [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]
        fn.decorator_list, fn.returns = [], None
        for a in fn.args.args + fn.args.kwonlyargs:
            a.annotation = None
        exec(compile(ast.Module(body=[fn], type_ignores=[]), f"{relpath}:{name}", "exec"), ns)
    return ns


from semabs_amd.synth import synth_ovssc_logits as ovssc_test_logits  # noqa: E402  (shared with tests/test_gpu_inference.py)


def g18_process_batch_ovssc():
    """visualize.process_batch_ovssc + get_sample_points (visualize.py:157-248, 283-298), EXECUTED: the functions are compiled from the
    reference's source and run with the reference's own TSDFVolume / check_pts_in_frustum / filter_pts_bounds; only the network call is a
    closed-form stand-in (`ovssc_test_logits`).  Pins sampling lattice, 2^k chunking incl. the ragged tail, TSDF at the sampling
    resolution, arg-max / cutoff / frustum / tsdf > 0 post-mask and the return form."""
    fusion, pc = refimport.load_reference_geometry()

    class Progress:                                            # rich.progress.Progress stand-in
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def add_task(self, *a, **k):
            return 0

        def update(self, *a, **k):
            pass

    # API drift (numpy 2, NEP 50): `xyz[:, 0] >= bounds[0, 0]` with an fp32 array and an np.float64 SCALAR is evaluated in f64 here but in fp32
    # under the reference's numpy 1.22 (value-based casting) - and fp32(-0.1) < -0.1, so the reference's own assertion (visualize.py:171)
    # would fail on its own lattice.  Give the comparison numpy 1.22's type: bounds in the points' dtype.
    fpb = lambda xyz, bounds: pc.filter_pts_bounds(xyz, np.asarray(bounds).astype(np.asarray(xyz).dtype))
    ns = {"np": np, "torch": torch, "TSDFVolume": fusion.TSDFVolume, "check_pts_in_frustum": pc.check_pts_in_frustum,
          "filter_pts_bounds": fpb, "Progress": Progress}
    _ref_functions("visualize.py", ["get_sample_points", "process_batch_ovssc"], ns)
    S, C = 40, 4
    sc = synth_scene(96, 96, seed=21)
    classes = [f"class{i}" for i in range(C)]

    def net(output_xyz_pts, **kw):                              # [1, 1, m, 3] -> [1, 1, m] for the class whose features were passed
        c = int(round(float(kw["input_feature_pts"].reshape(-1)[0])))
        return ovssc_test_logits(output_xyz_pts.reshape(-1, 3).float(), C)[c][None, None]

    n_in = 500
    batch = {"ovssc_obj_classes": classes, "input_xyz_pts": torch.zeros(n_in, 3),
             "input_feature_pts": torch.arange(C, dtype=torch.float32)[:, None].repeat(1, n_in),      # the stand-in reads the class id from here
             "rgb": sc["rgb"], "depth": sc["depth"], "cam_intr": sc["cam_intr"], "cam_extr": sc["cam_pose"]}
    # numba types the python-float voxel size as float64 (SURVEY.md 8c shim ii-b): the numpy-2 run must do the same
    orig_init = fusion.TSDFVolume.__init__

    def init64(self, vol_bnds, voxel_size):
        orig_init(self, vol_bnds, np.float64(voxel_size))
        self._voxel_size = np.float64(self._voxel_size)
    fusion.TSDFVolume.__init__ = init64
    orig_integrate = fusion.TSDFVolume.integrate
    fusion.TSDFVolume.integrate = lambda self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.0: orig_integrate(
        self, color_im, depth_im, cam_intr, cam_pose, np.float64(obs_weight))
    fusion.TSDFVolume.get_volume = lambda self: (self._tsdf_vol_cpu, None)        # NEP-50 overflow in the colour unpacking (shim ii); [0] is what the path reads
    try:
        vols = ns["process_batch_ovssc"](net=net, batch=batch, scene_bounds=SCENE_BOUNDS, device="cpu", num_input_pts=64,
                                         sampling_shape=(S, S, S), num_pts_per_pass=2 ** 14, cutoff=-3.0)
        pts = ns["get_sample_points"](sampling_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, device="cpu")
    finally:
        fusion.TSDFVolume.__init__, fusion.TSDFVolume.integrate = orig_init, orig_integrate
    assert list(vols.keys()) == classes and all(v.shape == (S, S, S) and v.dtype == np.float32 for v in vols.values())
    stack = np.stack([vols[c] for c in classes])
    assert set(np.unique(stack)) <= {0.0, 1.0}
    save("g18_process_batch_ovssc", meta=np.asarray([S, C, 96, 21, 2 ** 14]), packed=np.packbits(stack.astype(bool).reshape(C, -1), axis=1),
         counts=stack.reshape(C, -1).sum(1).astype(np.int64), points_sha=digest(pts.numpy()), points_sub=pts.numpy()[::997].copy())
    print("    per-class voxel counts", stack.reshape(C, -1).sum(1))


def g19_relevancy_storage():
    """generate_relevancy.generate_saliency_helper (:93-145) and SceneCompletionDataset.get_scene_patches (dataset.py:687-871), EXECUTED from
    the reference's source against in-memory stand-ins for ray / FileLock / h5py (containers only): what is written for a scene, and what the
    loader hands back for it (row pick, mean subtraction, bilinear resize); x 50 of dataset.py:1053 applied last."""
    rng = np.random.default_rng(190)
    res = {}
    for tag, (L, H, W, h, w) in {"a": (5, 96, 96, 40, 40), "b": (3, 120, 90, 64, 48), "c": (4, 60, 60, 128, 128)}.items():
        maps = torch.from_numpy((rng.standard_normal((L, H, W)) * 0.01).astype(np.float32))
        feats = torch.from_numpy(rng.standard_normal((L, 512)).astype(np.float32))
        labels = [f"label{i}" for i in range(L)]
        store = {"written": {}, "rows": None}

        class Remote:
            def __init__(self, fn):
                self.remote = fn

        class Wrapper:
            get_clip_saliency = Remote(lambda **kw: (maps.clone(), feats.clone()))

        class Ray:
            @staticmethod
            def get(x):
                return x

        class Lock:
            def __init__(self, *a):
                pass

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

        class Group(dict):
            def create_group(self, name):
                g = Group(); self[name] = g
                return g

        class DS:
            shape = (0, h, w)

        class File:
            def __init__(self, *a, **k):
                self.d = {"data": Group(), "saliencies": DS()}

            def __enter__(self):
                return self.d

            def __exit__(self, *a):
                return False

        H5 = type("H5", (), {"File": File, "regionref_dtype": "regionref"})

        def write_to_hdf5(group, key, value, dtype=None, replace=False):
            store["written"][key] = value

        def resize_and_add_data(dataset, data):
            store["rows"] = data
            return list(range(len(data)))

        ns = {"np": np, "torch": torch, "ray": Ray, "FileLock": Lock, "h5py": H5, "write_to_hdf5": write_to_hdf5,
              "resize_and_add_data": resize_and_add_data, "saliency_configs": {"ours": lambda img_dim: {}}, "imagenet_templates": []}
        _ref_functions("generate_relevancy.py", ["generate_saliency_helper"], ns)
        ns["generate_saliency_helper"](Wrapper(), {"rgb": np.zeros((H, W, 3), np.uint8)}, ["{}"], labels, "scene.hdf5", False)
        stored = store["rows"]
        tf = store["written"]["rgb|ours|saliency_text_label_features"]
        names = store["written"]["rgb|ours|saliency_text_labels"]
        # ---- loader -------------------------------------------------------------------------------------------------------------------
        keep = sorted(rng.choice(L, size=max(1, L - 1), replace=False).tolist())
        prefix = "data/saliencies/rgb|ours"

        class Arr:
            def __init__(self, shape):
                self.shape = shape

        sal = stored.numpy()
        file = {"data/objid_to_class": np.array([f"{labels[i]}[{i}]" for i in keep]).astype("S"),
                f"{prefix}|saliency_text_labels": names, "saliencies": sal, prefix: np.array([np.s_[i:i + 1] for i in range(len(sal))], dtype=object),
                f"{prefix}|saliency_text_label_features": tf.numpy(), "rgb": Arr((1, H, W, 3))}
        ns2 = {"np": np, "torch": torch, "synonyms": {}}
        _ref_functions("dataset.py", ["deref_h5py"], ns2)
        _ref_functions("dataset.py", ["get_scene_patches"], ns2, in_class="SceneCompletionDataset")
        sp = ns2["get_scene_patches"](file, -1, "rgb", "ours", False, True)
        assert list(sp["patch_labels"]) == [labels[i] for i in keep]
        res.update({f"{tag}_maps": maps.numpy(), f"{tag}_feats": feats.numpy(), f"{tag}_dims": np.asarray([L, H, W, h, w]),
                    f"{tag}_stored": stored.numpy(), f"{tag}_tf": tf.numpy(), f"{tag}_names": np.asarray(names).astype(str),
                    f"{tag}_rows": np.asarray(keep, np.int64), f"{tag}_loaded50": (sp["patch_saliencies"] * 50).numpy(),
                    f"{tag}_label_features": sp["patch_label_features"].numpy()})
    save("g19_relevancy_storage", **res)


if __name__ == "__main__" and "g18" in sys.argv[1:]:
    g18_process_batch_ovssc()
if __name__ == "__main__" and "g19" in sys.argv[1:]:
    g19_relevancy_storage()


# ---- appended (round 2): f4 - ViT-L/14, the true 13-layer rollout, through the unmodified reference ------------------------------
def g21_vit_l14():
    """ClipGradcam.forward / interpret of the reference (clip_gradcam.py:58-132) for "ViT-L/14" (24 blocks, blocks 11..23 enter the rollout,
    16 heads, 257 tokens) on 2 tiles x 3 labels, positive_attn_only True / False, seeded random-init weights of that architecture."""
    rc = refimport.load_reference_clip("ViT-L/14", seed=0)
    gc = rc.ClipWrapper.clip_gradcam
    assert gc.num_res_attn_blocks == 16 and len(list(gc.model.visual.transformer.resblocks.children())) == 24
    labels = ["chair", "table", "lamp"]
    gc.templates = [DEFAULT_PROMPT]
    gc.set_classes(labels)
    w_text = torch.cat([gc.class_to_language_feature[c] for c in labels], dim=1)
    tiles = _tiles_from_seed(rc, 2, seed=7)
    out = {"w_text": w_text.numpy(), "tiles_sum": np.float64(tiles.double().sum().item())}
    with torch.no_grad():
        out["feat"] = gc.model.encode_image(tiles).numpy()
    for pos in (True, False):
        gc.positive_attn_only = pos
        t = time.time()
        rel = gc(x=tiles, o=labels)
        print(f"    ViT-L/14 interpret pos={pos}: {time.time() - t:.1f}s  max|rel| {rel.abs().max():.4g}", flush=True)
        out[f"rel_pos{int(pos)}"] = rel.detach().numpy()
    feats = gc.model.encode_image(tiles)
    feats = feats / feats.norm(dim=-1, keepdim=True)
    out["logits"] = (100.0 * feats @ w_text).detach().numpy()
    save("g21_vit_l14", **out)


if __name__ == "__main__" and "g21" in sys.argv[1:]:
    g21_vit_l14()


# ---- appended (round 3): config 5 at its stated size (g22: 128^3, 4 descriptions, 80 000 / 400 000 points) and the reference's OWN gradient
# ---- spread under a 1e-6 relative weight perturbation (g22, and g20s for the 64^3 golden) ---------------------------------------------------
REL4 = [["behind"], ["on"], ["behind"], ["in"]]


def _vool_reference_step(S, N, M, D, rel_names, perturb_seed=None, eps=1e-6, step=True):
    """The unmodified reference: SemAbsVOOL forward -> BCE-with-logits -> backward -> clip_grad_norm_ -> arm.optim.lamb.Lamb.step (train_vool.py:118-178,
    utils.py:404-417).  perturb_seed: every parameter is first multiplied by (1 + eps * N(0, 1)) element-wise - the fp32-rounding-level perturbation
    whose effect on the reference's own gradients is the yardstick for the HIP path's deviation."""
    from semabs_amd.weights import make_semabsvool_state_dict
    net, _ = refimport.load_reference_net()
    if refimport.REF not in sys.path:
        sys.path.insert(0, refimport.REF)
    from arm.optim.lamb import Lamb
    m = net.SemAbsVOOL(pointing_method="cosine_sim", pointing_dim=64, device="cpu", decoder_concat_xyz_pts=True,
                       voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8,
                       unet_num_levels=6, network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128,
                       reduce_method="max", batch_size=1)
    sd = make_semabsvool_state_dict(seed=3)
    if perturb_seed is not None:
        gen = torch.Generator().manual_seed(perturb_seed)
        sd = {k: (v * (1 + eps * torch.randn(v.shape, generator=gen)) if torch.is_floating_point(v) and "steps" not in k else v) for k, v in sd.items()}
    m.load_state_dict(sd, strict=True)
    m.train()
    xyz, feat, q = _semabs_inputs(S, N, M, 2 * D, seed=13)
    label = (np.random.default_rng(131).random((1, D, M)) < 0.3).astype(np.float32)
    params = [(k, p) for k, p in m.named_parameters()]
    opt = Lamb([p for _, p in params], lr=1e-3, weight_decay=1e-5)
    before = {k: p.detach().clone() for k, p in params}
    t = time.time()
    out = m(output_xyz_pts=torch.from_numpy(q[:, :D]), spatial_relation_name=rel_names, input_xyz_pts=torch.from_numpy(xyz),
            input_target_saliency_pts=torch.from_numpy(feat[:, :D]), input_reference_saliency_pts=torch.from_numpy(feat[:, D:]), tsdf_vol=None)
    lab = torch.from_numpy(label)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, lab, weight=torch.ones_like(lab))
    opt.zero_grad()
    loss.backward()
    print(f"    reference fwd + bwd at {S}^3, {D} descriptions: {time.time() - t:.1f}s", flush=True)
    r = dict(label=label, loss=float(loss.item()), logits=out.detach().numpy().copy(), names=[k for k, _ in params],
             grads={k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in params})
    if step:
        r["total_norm"] = float(torch.nn.utils.clip_grad_norm_([p for _, p in params], 2.0))
        opt.step()
        r["new"] = {k: p.detach().numpy().copy() for k, p in params}
        r["before"] = {k: v.numpy() for k, v in before.items()}
    return r


def _spread(a, b):
    """Deviation of run b from run a in the statistics the GPU tests assert: per-tensor gradient norm (relative), relative L2 of each gradient
    tensor, total norm, loss, logits."""
    gtot = float(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in a["grads"].values() if g is not None)))
    names, norm_rel, l2_rel = [], [], []
    for k in a["names"]:
        ga, gb = a["grads"][k], b["grads"][k]
        if ga is None:
            continue
        na = float(np.linalg.norm(ga.astype(np.float64)))
        names.append(k)
        norm_rel.append(abs(float(np.linalg.norm(gb.astype(np.float64))) - na) / max(na, 1e-300))
        l2_rel.append(float(np.linalg.norm((gb - ga).astype(np.float64))) / max(na, 1e-300))
    return dict(spread_names=np.asarray(names), spread_norm_rel=np.asarray(norm_rel), spread_l2_rel=np.asarray(l2_rel), spread_gtot=np.float64(gtot),
                spread_loss_rel=np.float64(abs(b["loss"] - a["loss"]) / a["loss"]), spread_logits_linf=np.float64(np.abs(b["logits"] - a["logits"]).max()),
                spread_total_rel=np.float64(abs(b.get("total_norm", 0.0) - a.get("total_norm", 0.0)) / max(a.get("total_norm", 1.0), 1e-300)))


def g22_vool_train128(S=128, N=80000, M=400000, D=4, name="g22_vool_train128", perturbations=(1, 2, 3)):
    a = _vool_reference_step(S, N, M, D, REL4[:D] if D <= 4 else None)
    res = {"meta": np.asarray([S, N, M, D, 13, 3, 131], np.int32), "label_packed": np.packbits(a["label"].astype(np.uint8).reshape(-1)),
           "loss": np.float64(a["loss"]), "total_norm": np.float64(a["total_norm"]), "names": np.asarray(a["names"]),
           "rel_names": np.asarray([r[0] for r in REL4[:D]])}
    li = sample_idx(a["logits"].size, 16384)
    res["logit_idx"], res["logits_s"] = li, a["logits"].reshape(-1)[li]
    res["logits_sum"], res["logits_abs"] = np.float64(a["logits"].astype(np.float64).sum()), np.float64(np.abs(a["logits"].astype(np.float64)).sum())
    gnorm, has, dnorm = [], [], []
    for k in a["names"]:
        g = a["grads"][k]
        has.append(g is not None)
        gnorm.append(0.0 if g is None else float(np.linalg.norm(g.astype(np.float64))))
        dnorm.append(float(np.linalg.norm((a["new"][k] - a["before"][k]).astype(np.float64))))
        if g is not None:
            if g.size <= 4096:
                res["grad/" + k] = g
            else:
                si = sample_idx(g.size, 2048)
                res["gradidx/" + k], res["grads/" + k] = si, g.reshape(-1)[si]
    res["grad_norm"], res["has_grad"], res["delta_norm"] = np.asarray(gnorm, np.float64), np.asarray(has), np.asarray(dnorm, np.float64)
    sp = None
    for ps in perturbations:
        b = _vool_reference_step(S, N, M, D, REL4[:D], perturb_seed=ps)
        s = _spread(a, b)
        if sp is None:
            sp = s
        else:                                                   # worst over the perturbations, element-wise
            for k in ("spread_norm_rel", "spread_l2_rel"):
                sp[k] = np.maximum(sp[k], s[k])
            for k in ("spread_loss_rel", "spread_logits_linf", "spread_total_rel"):
                sp[k] = max(sp[k], s[k])
        del b
    res.update(sp)
    print(f"    {name}: loss {a['loss']:.6f}, total norm {a['total_norm']:.4e}; reference self-spread under 1e-6 perturbation: "
          f"grad-norm rel max {sp['spread_norm_rel'].max():.3e} median {np.median(sp['spread_norm_rel']):.3e}, "
          f"grad L2 rel max {sp['spread_l2_rel'].max():.3e} median {np.median(sp['spread_l2_rel']):.3e}, total {float(sp['spread_total_rel']):.3e}", flush=True)
    save(name, **res)


def g20s_spread64():
    """The 64^3 golden's companion: the reference's own gradient spread at g20's configuration (3 descriptions, 12 000 / 4 000 points)."""
    a = _vool_reference_step(64, 12000, 4000, 3, REL4[:3])
    g = np.load(os.path.join(HERE, "g20_vool_train64.npz"))
    assert abs(a["loss"] - float(g["loss"])) <= 1e-9 and np.array_equal(a["logits"], g["logits"]), "base run must reproduce g20"
    sp = None
    for ps in (1, 2, 3):
        s = _spread(a, _vool_reference_step(64, 12000, 4000, 3, REL4[:3], perturb_seed=ps))
        if sp is None:
            sp = s
        else:
            for k in ("spread_norm_rel", "spread_l2_rel"):
                sp[k] = np.maximum(sp[k], s[k])
            for k in ("spread_loss_rel", "spread_logits_linf", "spread_total_rel"):
                sp[k] = max(sp[k], s[k])
    print(f"    g20s: grad-norm rel max {sp['spread_norm_rel'].max():.3e} median {np.median(sp['spread_norm_rel']):.3e}, grad L2 rel max "
          f"{sp['spread_l2_rel'].max():.3e} median {np.median(sp['spread_l2_rel']):.3e}, total {float(sp['spread_total_rel']):.3e}", flush=True)
    save("g20s_vool_train64_spread", **sp)


if __name__ == "__main__" and "g20s" in sys.argv[1:]:
    g20s_spread64()
if __name__ == "__main__" and "g22" in sys.argv[1:]:
    g22_vool_train128()


# ---- appended (round 3): imagenet prompt ensemble (g24) and prep_data executed from the reference's source (g25) ------------------------------
def g24_prompt_ensemble():
    """`imagenet_prompt_ensemble=True` as generate_relevancy.py:70-80 does it: `prompts=imagenet_templates` (80 templates) into the unmodified
    `ClipWrapper.get_clip_saliency`; 2 labels, ViT-B/32, chefer_et_al at 96^2.  Also pins the template table itself (digest + count) and the
    reference tokenizer's ids for the 160 strings (the BPE table is not on the GPU box)."""
    rc = refimport.load_reference_clip("ViT-B/32", seed=0)
    import CLIP.clip.clip_explainability as rexp
    W = rc.ClipWrapper
    tpl = list(rc.imagenet_templates)
    labels = ["chair", "table"]
    texts = [t.format(c) for c in labels for t in tpl]
    img = synth_rgb(96, 96, seed=42)
    cfg = dict(rc.saliency_configs["chefer_et_al"](96), imagenet_prompt_ensemble=True)
    t = time.time()
    maps, feats = W.get_clip_saliency(img=img, text_labels=labels, prompts=tpl, **cfg)
    print(f"    prompt ensemble, 2 labels x {len(tpl)} templates: {time.time() - t:.1f}s  max|map| {maps.abs().max():.4g}")
    save("g24_prompt_ensemble", maps=maps.numpy(), text=feats.numpy(), tokens=rexp.tokenize(texts).numpy().astype(np.int32),
         templates_sha=np.frombuffer(hashlib.sha256("\n".join(tpl).encode()).digest(), dtype=np.uint8), n_templates=np.int32(len(tpl)),
         labels=np.asarray(labels))


def g25_prep_data():
    """visualize.prep_data (visualize.py:61-154) EXECUTED from the reference's source with the reference's own get_pointcloud / filter_pts_bounds;
    the CLIP call is a closed-form stand-in (`semabs_amd.synth.synth_relevancy`: a map per label string) shared with the GPU test, plotting /
    directory creation are no-ops.  Pins: relevancy key set, x 50, mean subtraction, in-bounds point selection and its order, the per-class /
    per-description feature stacks, the returned key set and the description strings."""
    import pickle
    import tempfile
    from semabs_amd.synth import synth_relevancy
    fusion, pc = refimport.load_reference_geometry()
    calls = []

    class ClipWrapper:                                          # stand-in: records what prep_data asks for
        @classmethod
        def get_clip_saliency(cls, img, text_labels, prompts, **kwargs):
            calls.append(dict(labels=[str(t) for t in text_labels], prompts=list(prompts), kwargs=dict(kwargs), shape=tuple(img.shape)))
            return torch.from_numpy(synth_relevancy(img, [str(t) for t in text_labels])), None

    class _Path:
        def __init__(self, p):
            pass

        def mkdir(self, **k):
            pass

    fpb = lambda xyz, bounds: pc.filter_pts_bounds(xyz, np.asarray(bounds).astype(np.float32))      # numpy-1.22 typing of the comparison (see g18)
    import CLIP.clip as rc
    ns = {"np": np, "torch": torch, "pickle": pickle, "os": os, "Path": _Path, "ClipWrapper": ClipWrapper, "saliency_configs": rc.saliency_configs,
          "get_pointcloud": pc.get_pointcloud, "filter_pts_bounds": fpb, "visualize_relevancies": lambda **k: None}
    _ref_functions("visualize.py", ["prep_data"], ns)
    sc = synth_scene(96, 96, seed=5)
    data = dict(rgb=sc["rgb"], depth=sc["depth"], cam_intr=sc["cam_intr"], cam_extr=sc["cam_pose"],
                ovssc_obj_classes=["chair", "table", "lamp"], descriptions=[("lamp", "on", "table"), ("cushion", "behind", "chair")])
    out = {"meta": np.asarray([96, 5], np.int32)}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scene_0007.pkl")
        pickle.dump(data, open(path, "wb"))
        for sub in (True, False):
            b = ns["prep_data"](data_pickle_path=path, scene_bounds=SCENE_BOUNDS, subtract_mean=sub, dump_path=d)
            tag = f"sub{int(sub)}"
            keys = calls[-1]["labels"]
            out[f"{tag}_keys"] = np.asarray(keys)
            out[f"{tag}_relevancies"] = b["relevancies"].numpy()
            out[f"{tag}_xyz_sha"] = digest(b["input_xyz_pts"].numpy())
            out[f"{tag}_xyz_sub"] = b["input_xyz_pts"].numpy()[::97].copy()
            out[f"{tag}_n"] = np.int64(len(b["input_xyz_pts"]))
            out[f"{tag}_rgb_sha"] = digest(b["input_rgb_pts"])
            for k in ("input_feature_pts", "input_target_saliency_pts", "input_reference_saliency_pts"):
                out[f"{tag}_{k}"] = b[k].numpy()
            out[f"{tag}_batch_keys"] = np.asarray(sorted(b.keys()))
            out[f"{tag}_descriptions"] = np.asarray(b["descriptions"])
            out[f"{tag}_relations"] = np.asarray(b["spatial_relation_name"])
            out[f"{tag}_scene_id"] = np.asarray(b["scene_id"])
            assert b["tsdf_vol"] is None and b["ovssc_obj_classes"] == data["ovssc_obj_classes"]
    c = calls[-1]
    out["call_prompts"] = np.asarray(c["prompts"])
    out["call_kwargs"] = np.asarray(sorted(c["kwargs"].keys()))
    assert c["kwargs"]["augmentations"] == 5 and c["kwargs"]["horizontal_flipping"] and c["shape"] == (96, 96, 3)
    save("g25_prep_data", **out)


if __name__ == "__main__" and "g24" in sys.argv[1:]:
    g24_prompt_ensemble()
if __name__ == "__main__" and "g25" in sys.argv[1:]:
    g25_prep_data()


# ---- appended (round 4): process_batch_vool executed from the reference's source (g26) --------------------------------------------------------
def g26_process_batch_vool():
    """visualize.process_batch_vool + get_sample_points (visualize.py:354-419, 283-298), EXECUTED: compiled from the reference's source and run with
    the reference's own filter_pts_bounds; the network call is the closed-form stand-in `semabs_amd.synth.synth_vool_logits`, which reads what the
    reference selected for each description (saliency rows by description index, `[[relation]]`).  Pins the sampling lattice, the 2^k chunking incl.
    the ragged tail, per-description selection, concatenation and the return form ({description: tensor of sampling_shape}, grid points)."""
    from semabs_amd.synth import synth_vool_logits
    fusion, pc = refimport.load_reference_geometry()

    class Progress:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def add_task(self, *a, **k):
            return 0

        def update(self, *a, **k):
            pass

    fpb = lambda xyz, bounds: pc.filter_pts_bounds(xyz, np.asarray(bounds).astype(np.asarray(xyz).dtype))      # numpy-1.22 typing (see g18)
    ns = {"np": np, "torch": torch, "filter_pts_bounds": fpb, "Progress": Progress}
    _ref_functions("visualize.py", ["get_sample_points", "process_batch_vool"], ns)
    S, n_in = 36, 300
    descs = [("lamp", "on", "table"), ("cushion", "behind", "chair"), ("mug", "on the left of", "lamp")]
    D = len(descs)
    rng = np.random.default_rng(26)
    tgt = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).repeat(1, n_in)         # constant along the points: whichever
    ref = torch.from_numpy(rng.standard_normal((D, 1)).astype(np.float32)).repeat(1, n_in)         # sub-sample is drawn, the first value is row d's
    seen = []

    def net(output_xyz_pts, spatial_relation_name, input_target_saliency_pts, input_reference_saliency_pts, **kw):
        # what the reference hands over: [None, None, [desc_idx], indices, None] on a [D, N] tensor -> (1, 1, num_input_pts, 1); [[relation]]
        assert tuple(input_target_saliency_pts.shape) == (1, 1, 64, 1) == tuple(input_reference_saliency_pts.shape)
        assert np.array(spatial_relation_name).shape == (1, 1) and tuple(output_xyz_pts.shape[:2]) == (1, 1) and tuple(kw["input_xyz_pts"].shape) == (64, 3)
        seen.append(int(output_xyz_pts.shape[-2]))
        v = synth_vool_logits(output_xyz_pts.reshape(-1, 3).float(), float(input_target_saliency_pts.reshape(-1)[0]),
                              float(input_reference_saliency_pts.reshape(-1)[0]), spatial_relation_name[0][0])
        return v[None, None]

    batch = {"descriptions": [f"the {a} {r} the {b}" for a, r, b in descs], "spatial_relation_name": [r for _, r, _ in descs],
             "input_xyz_pts": torch.zeros(n_in, 3), "input_target_saliency_pts": tgt, "input_reference_saliency_pts": ref}
    preds, pts = ns["process_batch_vool"](net=net, batch=batch, scene_bounds=SCENE_BOUNDS, device="cpu", num_input_pts=64,
                                          sampling_shape=(S, S, S), num_pts_per_pass=2 ** 13)
    assert list(preds.keys()) == batch["descriptions"] and all(tuple(v.shape) == (S, S, S) and v.dtype == torch.float32 for v in preds.values())
    stack = torch.stack([preds[d] for d in batch["descriptions"]]).numpy()
    save("g26_process_batch_vool", meta=np.asarray([S, D, n_in, 2 ** 13], np.int64), volumes=stack, tgt=tgt[:, 0].numpy(), ref=ref[:, 0].numpy(),
         relations=np.asarray(batch["spatial_relation_name"]), descriptions=np.asarray(batch["descriptions"]), points_sha=digest(pts.numpy()),
         chunks=np.asarray(seen[: len(seen) // D], np.int64))
    print("    chunks per description", seen[: len(seen) // D], " value range", float(stack.min()), float(stack.max()))


if __name__ == "__main__" and "g26" in sys.argv[1:]:
    g26_process_batch_vool()


# ---- appended (round 4): the reference's only REAL input, scene_files/arkit_vn_poster.pkl, through prep_data -> CLIP relevancy -> SemAbs3D (g27) ---
def g27_real_scene():
    """scene_files/arkit_vn_poster.pkl (256 x 192 RGB-D of an ARKit capture: non-square, real depth, 14 OVSSC classes + 3 descriptions -> 17 relevancy
    keys) through the reference end to end: `visualize.prep_data` compiled from its source and run with the UNMODIFIED `ClipWrapper.get_clip_saliency`
    (ViT-B/32, seeded weights; "ours" with augmentations = 0 so that no random colour jitter enters - everything else of the config as shipped),
    the reference's own get_pointcloud / filter_pts_bounds, then the reference's `SemAbs3D` (seeded weights, 128^3 - the released voxel grid) on all 14 classes
    with a fixed sub-sample and fixed query points.  The scene itself (a data file of the reference) travels in the fixture; so do the token ids of the 17
    prompts (the BPE table is not on the GPU box)."""
    import pickle
    import tempfile
    fusion, pc = refimport.load_reference_geometry()
    rc = refimport.load_reference_clip("ViT-B/32", seed=0)
    import CLIP.clip.clip_explainability as rexp
    net_mod, _ = refimport.load_reference_net()
    src = os.path.join(refimport.REF, "scene_files", "arkit_vn_poster.pkl")
    data = pickle.load(open(src, "rb"))
    calls = []

    class Wrapper:                                              # records the call, forwards it unchanged to the reference's ClipWrapper
        @classmethod
        def get_clip_saliency(cls, img, text_labels, prompts, **kwargs):
            calls.append(dict(labels=[str(t) for t in text_labels], prompts=list(prompts), kwargs=dict(kwargs)))
            return rc.ClipWrapper.get_clip_saliency(img=img, text_labels=text_labels, prompts=prompts, **kwargs)

    class _Path:
        def __init__(self, p):
            pass

        def mkdir(self, **k):
            pass

    fpb = lambda xyz, bounds: pc.filter_pts_bounds(xyz, np.asarray(bounds).astype(np.float32))      # numpy-1.22 typing of the comparison (see g18)
    cfgs = {"ours": lambda h: dict(rc.saliency_configs["ours"](h), augmentations=0)}
    ns = {"np": np, "torch": torch, "pickle": pickle, "os": os, "Path": _Path, "ClipWrapper": Wrapper, "saliency_configs": cfgs,
          "get_pointcloud": pc.get_pointcloud, "filter_pts_bounds": fpb, "visualize_relevancies": lambda **k: None}
    _ref_functions("visualize.py", ["prep_data"], ns)
    t = time.time()
    with tempfile.TemporaryDirectory() as d:
        b = ns["prep_data"](data_pickle_path=src, scene_bounds=SCENE_BOUNDS, subtract_mean=True, dump_path=d)
    keys = calls[-1]["labels"]
    print(f"    prep_data on the real scene: {time.time() - t:.1f}s, {len(keys)} relevancy keys, {len(b['input_xyz_pts'])} in-bounds points", flush=True)
    rel = b["relevancies"].numpy()                              # [17, 256, 192], x 50, mean-subtracted
    prompt = calls[-1]["prompts"][0]
    tokens = rexp.tokenize([prompt.format(k) for k in keys]).numpy().astype(np.int32)
    # ---- SemAbs3D (the reference's module, seeded weights) on four classes, fixed sub-sample / query points ----
    S, npts, M = 128, 40000, 4096
    m = net_mod.SemAbs3D(voxel_shape=(S, S, S), scene_bounds=SCENE_BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8, unet_num_levels=6,
                         network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128, reduce_method="max", output_dim=1,
                         device="cpu", decoder_concat_xyz_pts=True, batch_size=1)
    m.load_state_dict(make_semabs3d_state_dict(seed=3), strict=True)
    m.eval()
    rng = np.random.default_rng(27)
    n_in = len(b["input_xyz_pts"])
    idx = rng.integers(0, n_in, size=npts)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    q = (lo + (hi - lo) * rng.random((M, 3))).astype(np.float32)
    cls_idx = list(range(len(data["ovssc_obj_classes"])))          # all 14 classes, like visualize.ovssc_inference
    logits = []
    t = time.time()
    with torch.no_grad():
        for c in cls_idx:
            out = m(input_xyz_pts=b["input_xyz_pts"][None, idx].float(), input_feature_pts=b["input_feature_pts"][None, [c]][:, :, idx, None].float(),
                    tsdf_vol=None, output_xyz_pts=torch.from_numpy(q)[None, None])
            logits.append(out.reshape(-1).numpy())
    print(f"    SemAbs3D {S}^3 x {len(cls_idx)} classes: {time.time() - t:.1f}s", flush=True)
    rows = np.asarray([0, 37, 101, 128, 200, 255])
    save("g27_real_scene", rgb=data["rgb"], depth=data["depth"], cam_intr=np.asarray(data["cam_intr"]), cam_extr=np.asarray(data["cam_extr"]),
         ovssc_obj_classes=np.asarray(data["ovssc_obj_classes"]), descriptions=np.asarray([list(x) for x in data["descriptions"]]),
         keys=np.asarray(keys), prompt=np.asarray(prompt), tokens=tokens, rel_sub=rel[:, ::2, ::2].copy(), rel_rows_idx=rows, rel_rows=rel[:, rows, :].copy(),
         rel_absmax=np.abs(rel).reshape(len(keys), -1).max(1), n_in=np.int64(n_in), xyz_sha=digest(b["input_xyz_pts"].numpy()),
         xyz_sub=b["input_xyz_pts"].numpy()[::211].copy(), feat_sub=b["input_feature_pts"].numpy()[:, ::211].copy(),
         tgt_sub=b["input_target_saliency_pts"].numpy()[:, ::211].copy(), ref_sub=b["input_reference_saliency_pts"].numpy()[:, ::211].copy(),
         idx=idx.astype(np.int64), q=q, cls_idx=np.asarray(cls_idx), logits=np.stack(logits), meta=np.asarray([S, npts, M, 3], np.int64),
         cfg_keys=np.asarray(sorted(calls[-1]["kwargs"].keys())))


if __name__ == "__main__" and "g27" in sys.argv[1:]:
    g27_real_scene()


# ---- appended (round 5): the colour jitter of the augmentation copies (g28) ------------------------------------------------------------------
# ClipWrapper.jittering_transforms = torchvision.transforms.ColorJitter(0.6, 0.6, 0.6, 0.1) on the PIL image (CLIP/clip/__init__.py:55-57, 246-247).
# torchvision is not in this image; its PIL path (0.13.1 functional_pil: ImageEnhance.Brightness / Contrast / Color, and the uint8 hue rotation of
# convert("HSV")) is four Pillow calls, executed here with this image's Pillow on FIXED (order, factors): all 24 op orders on synth_rgb(480, 480),
# each op alone, and both HSV conversions on a 2^21-colour lattice (the generator also checks the oracle on all 2^24 colours before writing).
def _tv_jitter_pil(img_u8, order, factors):
    from PIL import Image, ImageEnhance
    im = Image.fromarray(img_u8)
    for op in order:
        f = float(factors[op])
        if op == 0:
            im = ImageEnhance.Brightness(im).enhance(f)
        elif op == 1:
            im = ImageEnhance.Contrast(im).enhance(f)
        elif op == 2:
            im = ImageEnhance.Color(im).enhance(f)
        else:                                                       # functional_pil.adjust_hue
            h, s, v = im.convert("HSV").split()
            np_h = np.array(h, dtype=np.uint8)
            np_h += np.uint8(int(f * 255) % 256)                    # np.uint8(hue_factor * 255) under the reference's numpy 1.22 (wraps when negative)
            im = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
    return np.array(im)


def g28_color_jitter():
    import itertools
    from PIL import Image
    import PIL
    from oracle import preprocess as opre
    print("g28 colour jitter (Pillow %s)" % PIL.__version__)
    img = synth_rgb(480, 480, seed=0)
    orders = np.asarray(list(itertools.permutations(range(4))), np.int32)                      # 24 x 4
    rng = np.random.default_rng(28)
    factors = np.stack([rng.uniform(0.4, 1.6, size=24), rng.uniform(0.4, 1.6, size=24), rng.uniform(0.4, 1.6, size=24), rng.uniform(-0.1, 0.1, size=24)], axis=1)
    factors[0] = [0.4, 1.6, 0.4, -0.1]; factors[1] = [1.6, 0.4, 1.6, 0.1]; factors[2] = [1.0, 1.0, 1.0, 0.0]      # range ends and the identity
    factors = factors.astype(np.float32).astype(np.float64)           # the C ABI carries fp32 factors; Pillow rounds its alpha to a C float anyway
    outs = [_tv_jitter_pil(img, list(o), f) for o, f in zip(orders, factors)]
    sha = np.stack([digest(o) for o in outs])
    sub = np.stack([o[::5, ::5] for o in outs])
    single_f = np.asarray([[0.55, 1.45], [0.62, 1.38], [0.47, 1.53], [-0.073, 0.091]])
    single_sha = np.stack([np.stack([digest(_tv_jitter_pil(img, [op], {op: f})) for f in single_f[op]]) for op in range(4)])
    # HSV conversions: lattice of 2^21 colours (every second value per channel) for the CPU test, all 2^24 checked against the oracle right here
    c = np.arange(1 << 24, dtype=np.uint32)
    allc = np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)
    hsv_all = np.array(Image.fromarray(allc).convert("HSV"))
    rgb_all = np.array(Image.fromarray(allc, "HSV").convert("RGB"))
    assert np.array_equal(opre.rgb_to_hsv_u8(allc), hsv_all) and np.array_equal(opre.hsv_to_rgb_u8(allc), rgb_all), "oracle != Pillow on the full colour cube"
    for o, f, ref in zip(orders, factors, outs):
        assert np.array_equal(opre.color_jitter(img, o, f), ref), (o, f)
    lat = allc.reshape(256, 256, 256, 3)[::2, ::2, ::2].reshape(-1, 1, 3)
    # every op alone on the full colour cube (4096 x 4096 image of all 2^24 colours), Pillow's bytes as sha256: the GPU test needs no CPU oracle time
    cube_ops = np.asarray([[0, 0.55], [0, 1.45], [1, 0.62], [1, 1.38], [2, 0.47], [2, 1.53], [3, 0.0], [3, 0.1], [3, -0.073]])
    cube_sha = []
    for opid, f in cube_ops:
        ref = _tv_jitter_pil(allc, [int(opid)], {int(opid): float(np.float32(f))})
        assert np.array_equal(opre.JITTER_OPS[int(opid)](allc, float(np.float32(f))), ref), (opid, f)
        cube_sha.append(digest(ref))
    save("g28_color_jitter", cube_ops=cube_ops, cube_sha=np.stack(cube_sha), orders=orders, factors=factors, sha=sha, sub=sub, single_f=single_f, single_sha=single_sha,
         lattice_hsv_sha=digest(np.array(Image.fromarray(np.ascontiguousarray(lat)).convert("HSV"))),
         lattice_rgb_sha=digest(np.array(Image.fromarray(np.ascontiguousarray(lat), "HSV").convert("RGB"))),
         allcolours_hsv_sha=digest(hsv_all), allcolours_rgb_sha=digest(rgb_all),
         meta=np.asarray([480, 480, 0, 1], np.int64), pillow=np.asarray(PIL.__version__))


if __name__ == "__main__" and "g28" in sys.argv[1:]:
    g28_color_jitter()


# ---- appended (round 6): the relevancy path on weights with TRAINED-checkpoint statistics ------------------------------------
def g29_trained_stats(which=("b32", "b16", "e2e", "head")):
    """`make_clip_state_dict(stats="trained")` (massive-activation channels, per-row DC offsets of ~4 sigma, peaked softmax; tools/clip_stats.py prints the
    statistics) loaded into the UNMODIFIED reference, text tower included:
      g29_vit_{b32,b16}: the g3g4 per-tile form - `ClipGradcam.forward` + `interpret` (autograd) on 3 tiles x 4 labels, positive_attn_only True / False,
                         zero-shot weights from the reference's own tokenizer + text tower, token ids stored (the GPU box has no BPE table);
      g29_e2e:           `get_clip_saliency`, "ours" with augmentations = 0, 4 labels: ViT-B/32 at 120 x 120 (positive_attn_only True and False), ViT-B/16 at 240 x 240;
      g29_headline_aug0: the BASELINE shape (480 x 480, ViT-B/16, 16 labels, "ours", augmentations = 0), stored like g16."""
    labels = ["chair", "table", "lamp", "sofa"]
    for arch, tag in (("ViT-B/32", "b32"), ("ViT-B/16", "b16")):
        if tag not in which:
            continue
        rc = refimport.load_reference_clip(arch, seed=0, stats="trained")
        import CLIP.clip.clip_explainability as rexp
        gc = rc.ClipWrapper.clip_gradcam
        gc.templates = [DEFAULT_PROMPT]
        gc.set_classes(labels)
        w_text = torch.cat([gc.class_to_language_feature[c] for c in labels], dim=1)
        tiles = _tiles_from_seed(rc, 3, seed=7)
        out = {"w_text": w_text.numpy(), "tokens": rexp.tokenize([DEFAULT_PROMPT.format(c) for c in labels]).numpy(), "labels": np.asarray(labels),
               "tiles_sum": np.float64(tiles.double().sum().item())}
        with torch.no_grad():
            out["feat"] = gc.model.encode_image(tiles).numpy()
        for pos in (True, False):
            gc.positive_attn_only = pos
            out[f"rel_pos{int(pos)}"] = gc(x=tiles, o=labels).detach().numpy()
        blk = list(gc.model.visual.transformer.resblocks.children())[-1]
        T = blk.attn_probs.shape[-1]
        out["probs_cls"] = blk.attn_probs.detach().view(3, 12, T, T)[:, :, 0, :].numpy()
        feats = gc.model.encode_image(tiles)
        feats = feats / feats.norm(dim=-1, keepdim=True)
        out["logits"] = (100.0 * feats @ w_text).detach().numpy()
        save(f"g29_vit_{tag}", **out)
    if "e2e" in which:
        out = {}
        for arch, name, H, pos in (("ViT-B/32", "b32_ours120", 120, True), ("ViT-B/32", "b32_ours120_signed", 120, False), ("ViT-B/16", "b16_ours240", 240, True)):
            rc = refimport.load_reference_clip(arch, seed=0, stats="trained")
            cfg = dict(rc.saliency_configs["ours"](H), augmentations=0, positive_attn_only=pos)
            img = synth_rgb(H, H, seed=42)
            t = time.time()
            maps, feats = rc.ClipWrapper.get_clip_saliency(img=img, text_labels=labels, prompts=[DEFAULT_PROMPT], **cfg)
            print(f"    g29 e2e {name}: {time.time() - t:.1f}s  max|map| {maps.abs().max():.4g}", flush=True)
            out[f"{name}_maps"] = maps.numpy()
            out[f"{name}_text"] = feats.numpy()
        save("g29_e2e", **out)
    if "head" in which:
        from semabs_amd.weights import DEFAULT_LABELS
        rc = refimport.load_reference_clip("ViT-B/16", seed=0, stats="trained")
        lab16 = list(DEFAULT_LABELS[:16])
        img = synth_rgb(480, 480, seed=0)
        cfg = dict(rc.saliency_configs["ours"](480), augmentations=0)
        t = time.time()
        maps, feats = rc.ClipWrapper.get_clip_saliency(img=img, text_labels=lab16, prompts=[DEFAULT_PROMPT], **cfg)
        dt = time.time() - t
        m = maps.numpy()
        print(f"    g29 headline aug0 (trained statistics): {dt:.1f}s  max|map| {np.abs(m).max():.5g}", flush=True)
        rows = np.asarray([0, 61, 122, 183, 244, 305, 366, 479])
        save("g29_headline_aug0", sub=m[:, ::4, ::4].copy(), rows_idx=rows, rows=m[:, rows, :].copy(), absmax=np.abs(m).reshape(16, -1).max(1),
             sums=m.astype(np.float64).reshape(16, -1).sum(1), sha=np.stack([digest(m[l]) for l in range(16)]), text=feats.numpy(),
             labels=np.asarray(lab16), seconds=np.float64(dt), cores=np.int64(os.cpu_count()))


if __name__ == "__main__" and any(a.startswith("g29") for a in sys.argv[1:]):
    g29_trained_stats(tuple(a.split(":")[1] for a in sys.argv[1:] if a.startswith("g29:")) or ("b32", "b16", "e2e", "head"))


if __name__ == "__main__" and "g30" in sys.argv[1:]:
    g9_semabs3d(stats="trained", name="g30_semabs3d_trained")
