"""GPU, BASELINE.json's full sizes (480 x 480, 16 labels, ViT-B/16 "ours", 128^3): the oracle cannot finish these in seconds, so the checks are
size-independent properties of the path - label-permutation equivariance of the relevancy stage, batch invariance of the UNet,
bit-exact integer geometry against closed forms, determinism run to run."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from semabs_amd.synth import SCENE_BOUNDS, synth_scene

pytestmark = pytest.mark.gpu

IMG, L, S = 480, 16, 128


@pytest.fixture(scope="module")
def pipe():
    from semabs_amd.scene import build_default
    return build_default("ViT-B/16", precision="exact", chunk_tiles=220, max_labels=L, voxel=S, text_tower=False)


def _w(seed=0):
    w = np.random.default_rng(seed).standard_normal((L, 512)).astype(np.float32)
    return w / np.linalg.norm(w, axis=1, keepdims=True)


def test_relevancy_label_permutation_equivariance_and_determinism(pipe):
    """2 448 tile forwards x 16 labels: a second run reproduces the first bit for bit (no atomics on this stage), and permuting the labels
    permutes the maps (the ViT forward is label-independent and each label's rollout touches only its own rows; rows of different labels
    share the batched fp16-operand VJP GEMMs, so the match is at that noise level - measured 8e-5 of the maximum - not to the bit)."""
    from semabs_amd.clip import ClipWrapper, saliency_configs
    cfg = saliency_configs["ours"](IMG)
    sc = synth_scene(IMG, IMG, seed=3)
    images = ClipWrapper.make_images(sc["rgb"], 0)                    # augmentations=0: no random jitter in a parity-style check
    w = _w()
    perm = np.random.default_rng(1).permutation(L)
    run = lambda ww: ClipWrapper.relevancy_device(images, torch.from_numpy(ww).cuda(), cfg["cropping_augmentations"], True, True)
    a, b, c = run(w), run(w[perm]), run(w)
    torch.cuda.synchronize()
    assert tuple(a.shape) == (L, IMG, IMG) and torch.isfinite(a).all()
    assert torch.equal(a, c)
    d = float((a[torch.from_numpy(perm).cuda()] - b).abs().max())
    print(f"label permutation: max |diff| {d:.3e} of max {float(a.abs().max()):.3e}")
    assert d <= 3e-4 * float(a.abs().max())
    assert float(a.abs().max()) > 0


def test_unet128_batch_invariance(pipe):
    """A volume's features do not depend on which other volumes share the launch (GroupNorm statistics are per volume; the only run-to-run
    freedom is the order of the fp64 statistic atomics)."""
    u = pipe.net.vol_feature_extractor
    rng = np.random.default_rng(5)
    x = torch.zeros(3, S, S, S, 16, device="cuda")
    occ = torch.from_numpy(rng.random((3, S, S, S)) < 0.03).cuda()
    x[occ] = torch.from_numpy(rng.standard_normal((int(occ.sum()), 16)).astype(np.float32)).cuda()
    full = u.forward_cl(x)
    solo = u.forward_cl(x[1:2].contiguous())
    torch.cuda.synchronize()
    scale = float(full[1].abs().max())
    assert float((full[1] - solo[0]).abs().max()) <= 1e-5 * scale
    assert torch.isfinite(full).all() and scale > 0


def test_geometry_full_size_closed_forms(pipe):
    """480 x 480 unprojection + voxel indices at 128^3: every index in range, identical to the fp32 closed form evaluated with torch on the
    same points, and the in-bounds mask equals the inclusive box test."""
    from semabs_amd.point_cloud import pointcloud_device
    sc = synth_scene(IMG, IMG, seed=11)
    bounds = np.array([SCENE_BOUNDS[0], SCENE_BOUNDS[1]], np.float64)
    xyz, mask = pointcloud_device(torch.from_numpy(sc["depth"]).cuda(), sc["cam_intr"], sc["cam_pose"], bounds)
    lo = torch.tensor(SCENE_BOUNDS[0], dtype=torch.float32, device="cuda"); hi = torch.tensor(SCENE_BOUNDS[1], dtype=torch.float32, device="cuda")
    inside = ((xyz >= lo) & (xyz <= hi)).all(dim=-1).view(-1)
    assert torch.equal(mask.view(-1).bool(), inside)
    assert 0.3 < float(inside.float().mean()) <= 1.0                 # the synthetic pose looks into the scene box
    vg = pipe.net.vg
    pts = xyz.view(-1, 3)[inside]
    flat = vg.flat_idxs(pts)
    off = torch.from_numpy(vg.offsets).cuda(); scl = torch.from_numpy(vg.scales).cuda()
    idx = ((pts + off) * scl).to(torch.int64).clamp_(0, S - 1)       # net.py:91-133: trunc of (p - lc) * (S - 1) / (uc - lc), fp32
    ref = idx[:, 0] * (S * S) + idx[:, 1] * S + idx[:, 2]
    assert torch.equal(flat, ref)
    assert int(flat.min()) >= 0 and int(flat.max()) < S ** 3
