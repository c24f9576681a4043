"""GPU: edge cases of the path against the oracle — non-square images (a scale with no tile), label chunking, injected
augmentation images, tile sharding, colliding / single / out-of-bounds points, empty query sets, invalid depth."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import geometry as og
from oracle import relevancy as orl
from oracle import semabs3d as os3
from semabs_amd.synth import SCENE_BOUNDS, synth_rgb, synth_scene
from semabs_amd.weights import make_clip_state_dict, make_semabs3d_state_dict

pytestmark = pytest.mark.gpu
BOUNDS = [list(SCENE_BOUNDS[0]), list(SCENE_BOUNDS[1])]


def _clip(max_labels=4, chunk=32):
    from semabs_amd.clip import ClipWrapper
    ClipWrapper.engine = None
    ClipWrapper("ViT-B/32", state_dict=make_clip_state_dict("ViT-B/32", 0, text_tower=False), chunk_tiles=chunk, max_labels=max_labels)
    return ClipWrapper


def _w(L, seed=0):
    w = np.random.default_rng(seed).standard_normal((L, 512)).astype(np.float32)
    return w / np.linalg.norm(w, axis=1, keepdims=True)


def _check(maps, ref, tol=5.5e-3):          # relative L-infinity; 3 x the largest value measured on these small shapes
    err = np.abs(maps - ref).max()
    assert err <= tol * np.abs(ref).max() and err <= 1e-3, (err, np.abs(ref).max())


def test_non_square_image_with_an_empty_scale():
    """64 x 48 image, "ours"(64): the 64-pixel scale produces no tile (wider than the image) and contributes 0 / 1e-5 = 0;
    the reference's row/column loop-bound quirk is exercised too."""
    from semabs_amd.clip import saliency_configs
    CW = _clip()
    H, W, L = 64, 48, 3
    img = synth_rgb(H, W, seed=9)
    cfg = dict(saliency_configs["ours"](H), augmentations=0)
    w = _w(L)
    maps = CW.relevancy_device(torch.from_numpy(img).cuda()[None].contiguous(), torch.from_numpy(w).cuda(), cfg["cropping_augmentations"],
                               True, True).cpu().numpy()
    sd = make_clip_state_dict("ViT-B/32", 0, text_tower=False)
    with torch.no_grad():
        ref = orl.relevancy_maps(sd, [img], torch.from_numpy(w).T.contiguous(), **cfg).numpy()
    assert maps.shape == (L, H, W)
    _check(maps, ref)


def test_label_chunking_injected_augmentations_and_tile_sharding():
    from semabs_amd.clip import saliency_configs
    from semabs_amd.dist import shard_range
    CW = _clip(max_labels=2, chunk=16)                     # 5 labels -> 3 label chunks; 2 images x 26 tiles -> 4 tile chunks
    H, L = 64, 5
    img = synth_rgb(H, H, seed=3)
    jit = synth_rgb(H, H, seed=4)                          # stands in for a colour-jittered copy
    cfg = dict(saliency_configs["chefer_et_al"](H), horizontal_flipping=True, augmentations=1,
               cropping_augmentations=[{"tile_size": 64, "stride": 16}, {"tile_size": 32, "stride": 8}])
    w = torch.from_numpy(_w(L, 2)).cuda()
    images = CW.make_images(img, 1, jittered_images=[jit])
    maps = CW.relevancy_device(images, w, cfg["cropping_augmentations"], True, True)
    sd = make_clip_state_dict("ViT-B/32", 0, text_tower=False)
    with torch.no_grad():
        ref = orl.relevancy_maps(sd, [img, jit], w.cpu().T.contiguous(), **cfg).numpy()
    _check(maps.cpu().numpy(), ref)
    # tile sharding: two "ranks" run disjoint tile slices; their concatenated per-tile relevances aggregate to the same maps
    rel_full, table, scales = CW.relevancy_device(images, w, cfg["cropping_augmentations"], True, True, return_tiles=True)
    parts = []
    for r in range(2):
        parts.append(CW.relevancy_device(images, w, cfg["cropping_augmentations"], True, True, tile_range=shard_range(len(table), r, 2),
                                         return_tiles=True)[0])
    summed = [torch.cat([parts[0][p], parts[1][p]], dim=1).contiguous() for p in range(2)]
    assert all(torch.equal(a, b) for a, b in zip(summed, rel_full))
    assert torch.equal(CW.aggregate_device(summed, scales, 2, H, H), maps)


def _net(S=16, precision="exact"):
    from semabs_amd.net import SemAbs3D
    m = SemAbs3D(voxel_shape=(S, S, S), scene_bounds=BOUNDS, unet_num_channels=16, unet_f_maps=16, unet_num_groups=8, unet_num_levels=4,
                 network_inputs=["saliency"], use_pts_feat_extractor=True, pts_feat_extractor_hidden_dim=128, reduce_method="max",
                 output_dim=1, device="cuda", decoder_concat_xyz_pts=True, batch_size=1, precision=precision)
    sd = make_semabs3d_state_dict(seed=5, unet_num_levels=4)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("case", ["all_in_one_voxel", "single_point", "points_outside_bounds"])
def test_semabs3d_degenerate_point_sets(case):
    """Collisions (every point in one voxel: a 300-long list, > the sorted-list fast path), a single point, and points far
    outside the bounds (clamped to the border voxels like the reference)."""
    S, P, M = 16, 2, 500
    m, sd = _net(S)
    rng = np.random.default_rng(1)
    if case == "all_in_one_voxel":
        xyz = (np.array([0.1, -0.2, 0.9]) + 1e-4 * rng.random((300, 3))).astype(np.float32)
    elif case == "single_point":
        xyz = np.array([[0.3, 0.3, 0.3]], np.float32)
    else:
        xyz = (rng.standard_normal((200, 3)) * 5).astype(np.float32)
    N = len(xyz)
    feat = rng.standard_normal((1, P, N, 1)).astype(np.float32)
    q = (np.array(BOUNDS[0]) + (np.array(BOUNDS[1]) - np.array(BOUNDS[0])) * rng.random((1, P, M, 3))).astype(np.float32)
    out = m.forward(torch.from_numpy(xyz[None]), torch.from_numpy(feat), None, torch.from_numpy(q)).cpu().numpy()
    with torch.no_grad():
        ref = os3.semabs3d_forward(sd, torch.from_numpy(xyz[None]), torch.from_numpy(feat), torch.from_numpy(q), BOUNDS, (S, S, S), num_levels=4).numpy()
    assert np.abs(out - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()


def test_empty_inputs_are_no_ops():
    from semabs_amd.net import VirtualGrid
    m, _ = _net(16)
    f = torch.zeros(2, 16, 16, 16, 16, dtype=m.vol_feature_extractor.act_dtype, device="cuda")
    assert m.decode(f, torch.zeros(2, 0, 3, device="cuda")).shape == (2, 0)             # no query points
    vg = VirtualGrid(np.array(BOUNDS), (16, 16, 16), batch_size=1)
    assert vg.flat_idxs(torch.zeros(0, 3, device="cuda")).numel() == 0


def test_tsdf_invalid_depth_and_behind_camera():
    """depth == 0 pixels are never integrated; voxels behind the camera (z <= 0) are outside the frustum."""
    from semabs_amd.fusion import TSDFVolume
    sc = synth_scene(64, 64, seed=2)
    depth = sc["depth"].copy()
    depth[::2] = 0.0                                        # invalid rows
    pose = sc["cam_pose"].copy()
    pose[0, 3] = 0.0                                        # camera inside the volume: half the voxels are behind it
    S = 32
    vs = (BOUNDS[1][0] - BOUNDS[0][0]) / S
    tv = TSDFVolume(np.array(BOUNDS).T.copy(), vs)
    tv.integrate(sc["rgb"], depth, sc["cam_intr"], pose, keep_pix=True)
    ref = og.TSDFVolume(np.array(BOUNDS).T, vs)
    ref.integrate(sc["rgb"], depth, sc["cam_intr"], pose)
    assert np.array_equal(tv._tsdf_vol_cpu, ref._tsdf_vol_cpu) and np.array_equal(tv._weight_vol_cpu, ref._weight_vol_cpu)
    assert np.array_equal(tv._color_vol_cpu, ref._color_vol_cpu)
    assert 0 < (ref._weight_vol_cpu > 0).sum() < ref._weight_vol_cpu.size
