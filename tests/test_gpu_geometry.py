"""GPU parity: HIP geometry kernels (through the C ABI) vs the oracle and the committed golden vectors.
Integer outputs (voxel indices, pixel indices) and the fp32 volumes they drive are compared bit-exactly."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from conftest import sha
from oracle import geometry as og
from semabs_amd.synth import synth_scene

pytestmark = pytest.mark.gpu
SCENE_BOUNDS = [[-1.0, -1.0, -0.1], [1.0, 1.0, 1.9]]


@pytest.mark.parametrize("tag,hw,S", [("48", 48, 32), ("480", 480, 128)])
def test_pointcloud_and_voxel_index(golden, tag, hw, S):
    from semabs_amd import point_cloud as pc
    from semabs_amd.net import VirtualGrid
    g = golden("g8_geometry")
    sc = synth_scene(hw, hw, seed=5)
    depth = torch.from_numpy(sc["depth"]).cuda()
    xyz, mask = pc.pointcloud_device(depth, sc["cam_intr"], sc["cam_pose"], np.array(SCENE_BOUNDS))
    ref = og.get_pointcloud(sc["depth"], sc["cam_intr"], sc["cam_pose"]).astype(np.float32)
    got = xyz.cpu().numpy()
    nbad = int((got != ref).sum())
    assert nbad == 0, f"{nbad} fp32 coordinates differ from the oracle"
    assert np.array_equal(sha(got), g[f"{tag}_pts32_sha"])
    assert np.array_equal(sha(mask.cpu().numpy().astype(bool)), g[f"{tag}_mask_sha"])
    vg = VirtualGrid(np.array(SCENE_BOUNDS), (S, S, S), batch_size=1)
    flat = vg.flat_idxs(xyz).cpu().numpy()
    assert np.array_equal(sha(flat.astype(np.int64)), g[f"{tag}_flat_sha"])
    idx3 = vg.get_points_grid_idxs(xyz)
    assert torch.equal(vg.flatten_idxs(idx3).cpu(), torch.from_numpy(flat))
    # numpy-facing drop-ins
    pts_np, _ = pc.get_pointcloud(sc["depth"], None, sc["cam_intr"], sc["cam_pose"])
    assert pts_np.dtype == np.float64 and np.array_equal(pts_np.astype(np.float32), ref)
    # true f64 results (not widened fp32): against the reference's own f64 points - equal up to the last bit of the 3-term dot product
    # (numpy's dgemm may fuse differently): |diff| <= 2 ulp of the largest coordinate
    ref64 = g["48_pts64"] if tag == "48" else g["480_pts64_s"]
    got64 = pts_np if tag == "48" else pts_np[g["480_si"]]
    assert np.abs(got64 - ref64).max() <= 2 * np.spacing(np.abs(ref64).max())
    assert np.abs(pts_np - pts_np.astype(np.float32)).max() > 1e-9          # genuinely carries more than fp32
    fr = pc.check_pts_in_frustum(ref[::3].astype(np.float64) * 1.01, sc["depth"], sc["cam_pose"], sc["cam_intr"])
    assert np.array_equal(sha(fr), g[f"{tag}_frustum_sha"])


def test_voxel_index_edge_cases():
    from semabs_amd.net import VirtualGrid
    vg = VirtualGrid(np.array(SCENE_BOUNDS), (128, 128, 128), batch_size=1)
    pts = np.array([[-1, -1, -0.1], [1, 1, 1.9], [-5, 7, 0.3], [0.99999994, -1.0000001, 1.8999999],
                    [-1 + 2 * 2 / 127, -1 + 2 * 64 / 127, -0.1 + 2 * 126 / 127]], np.float32)
    got = vg.flat_idxs(torch.from_numpy(pts).cuda()).cpu().numpy()
    ref = og.flatten_idxs(og.points_grid_idxs(pts, SCENE_BOUNDS, (128, 128, 128)), (128, 128, 128))
    assert np.array_equal(got, ref)
    assert vg.flat_idxs(torch.zeros(0, 3).cuda()).numel() == 0      # empty input


@pytest.mark.parametrize("tag,hw,S", [("16", 48, 16), ("32", 64, 32), ("128", 480, 128)])
def test_tsdf(golden, tag, hw, S):
    from semabs_amd.fusion import TSDFVolume
    g = golden("g11_tsdf")
    vs = (SCENE_BOUNDS[1][0] - SCENE_BOUNDS[0][0]) / S
    tv = TSDFVolume(np.array(SCENE_BOUNDS).T.copy(), vs)
    assert np.array_equal(tv._vol_dim, g[f"{tag}_dim"])
    for k in range(2 if S <= 32 else 1):
        sc = synth_scene(hw, hw, seed=6 if k == 0 else 16)
        tv.integrate(sc["rgb"], sc["depth"], sc["cam_intr"], sc["cam_pose"], keep_pix=True)
    assert np.array_equal(sha(tv.last_pix.cpu().numpy()), g[f"{tag}_pix_sha"])      # int64 pixel indices
    assert np.array_equal(sha(tv._weight_vol_cpu), g[f"{tag}_weight_sha"])
    assert np.array_equal(sha(tv._tsdf_vol_cpu), g[f"{tag}_tsdf_sha"])
    assert np.array_equal(sha(tv._color_vol_cpu), g[f"{tag}_color_sha"])
    tsdf, col = tv.get_volume()
    assert tsdf.shape == (S, S, S) and col.shape == (3, S, S, S) and col.dtype == np.uint8


def test_compact_subsample_matches_nonzero_and_the_oracle_draw():
    """semabs_compact_subsample: pix = ascending in-bounds pixel ids (== torch.nonzero), n_in on the device, sel = the counter-based seeded
    draw with replacement the oracle restates (oracle/scene.py:subsample_indices) - incl. an empty mask and a single survivor."""
    from oracle.scene import subsample_indices
    from semabs_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for n, p, num, seed in [(480 * 480, 0.37, 80000, 7), (1000, 0.5, 333, 0), (1025, 1.0, 64, 123456789), (5000, 0.0, 16, 1), (77, 0.02, 40, 2 ** 40 + 5)]:
        m = (rng.random(n) < p).astype(np.uint8)
        if p > 0 and m.sum() == 0:
            m[n // 2] = 1
        mask = torch.from_numpy(m).to(dev)
        pix = torch.full((n,), -1, dtype=torch.int64, device=dev)
        n_in = torch.zeros(1, dtype=torch.int64, device=dev)
        sel = torch.full((num,), -1, dtype=torch.int64, device=dev)
        _lib.call("semabs_compact_subsample", _lib.ptr(mask), n, seed, num, _lib.ptr(pix), _lib.ptr(n_in), _lib.ptr(sel), _lib.stream())
        ref_pix = np.nonzero(m)[0]
        assert int(n_in.item()) == len(ref_pix)
        assert np.array_equal(pix.cpu().numpy()[: len(ref_pix)], ref_pix)
        if len(ref_pix):
            assert np.array_equal(sel.cpu().numpy(), ref_pix[subsample_indices(seed, len(ref_pix), num)])
        else:
            assert not sel.cpu().numpy().any()
