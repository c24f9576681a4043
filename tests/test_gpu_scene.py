"""GPU end-to-end: uint8 RGB + depth -> relevancy -> point features -> SemAbs3D logits -> masked labels, the whole HIP
pipeline against the oracle's run of the same recipe (small shapes the oracle finishes in seconds)."""
import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import scene as osc
from semabs_amd.synth import SCENE_BOUNDS, synth_scene
from semabs_amd.weights import make_clip_state_dict, make_semabs3d_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["exact", "fp16"])
def test_scene_end_to_end(precision, bar):
    from semabs_amd.clip import saliency_configs
    from semabs_amd.scene import build_default
    S, H, L, npts = 32, 96, 4, 4000
    pipe = build_default("ViT-B/32", precision=precision, chunk_tiles=64, max_labels=4, voxel=S, text_tower=False, num_input_pts=npts,
                         config="chefer_et_al")
    sc = synth_scene(H, H, seed=8)
    rng = np.random.default_rng(0)
    w = rng.standard_normal((L, 512)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    res = pipe.run(pipe.upload(sc), torch.from_numpy(w).cuda(), seed=5)
    ref = osc.run_scene(make_clip_state_dict("ViT-B/32", 0, text_tower=False), make_semabs3d_state_dict(seed=3), sc, torch.from_numpy(w),
                        [list(SCENE_BOUNDS[0]), list(SCENE_BOUNDS[1])], S, npts, 5, saliency_configs["chefer_et_al"](H))
    assert res.n_in_bounds == ref["n_in_bounds"]
    # relevancy maps (x 50): absolute tolerance 50 * 1e-3 would be the BASELINE bar; we hold 1% of the max
    rel = res.relevancies.cpu().numpy() * 50
    rel = rel - rel.mean(axis=0, keepdims=True)
    r_ref = ref["relevancies"].numpy()
    e_rel = np.abs(rel - r_ref).max() / np.abs(r_ref).max()
    print(f"scene relevancies (x 50, mean-subtracted): relative L-inf {e_rel:.2e}")
    assert bar("relevancy_rel", e_rel, 1e-2)             # mean subtraction shrinks the reference's range; measured 3.74e-3 (tests/golden/measured_errors.json)
    lg, lg_ref = res.logits.cpu().numpy(), ref["logits"].numpy()
    err = np.abs(lg - lg_ref).max()
    print(f"{precision}: logits Linf {err:.3e} (max|ref| {np.abs(lg_ref).max():.3f})")
    # the point features inherit the fp16-GEMM relevancy error (~1e-3 relative); logits are O(0.5)
    assert bar("logits", err, 5e-3 if precision == "exact" else 2e-2)      # measured 5.6e-4 / 1.6e-3
    assert torch.equal(res.tsdf.cpu(), torch.from_numpy(ref["tsdf"]))                      # TSDF volume bit-exact
    lab, lab_ref = res.labels.cpu().numpy(), ref["labels"]
    assert ((lab == -1) == (lab_ref == -1)).mean() > 0.999
    both = (lab >= 0) & (lab_ref >= 0)
    assert (lab[both] == lab_ref[both]).mean() > 0.97                                      # argmax flips only on near-ties


def test_scene_with_no_point_in_bounds_is_loud():
    """The reference fails on an empty in-bounds cloud (np.random.choice on an empty population, visualize.py:193).  The device path draws the
    sub-sample without a host synchronisation, so it must poison the result instead of returning the UNet's answer for 4 000 copies of pixel 0
    (ADVICE round 3): NaN logits, label -1 everywhere, and `n_in_bounds` raises on first read."""
    from semabs_amd.scene import build_default
    S, H, L, npts = 32, 96, 2, 4000
    pipe = build_default("ViT-B/32", precision="fp16", chunk_tiles=64, max_labels=L, voxel=S, text_tower=False, num_input_pts=npts, config="chefer_et_al")
    sc = synth_scene(H, H, seed=8)
    sc["depth"] = np.full_like(sc["depth"], 50.0)                                          # every pixel far outside scene_bounds
    w = torch.randn(L, 512, generator=torch.Generator().manual_seed(0))
    res = pipe.run(pipe.upload(sc), (w / w.norm(dim=1, keepdim=True)).cuda(), seed=1)
    assert bool(torch.isnan(res.logits).all())
    assert bool((res.labels == -1).all())
    with pytest.raises(RuntimeError, match="scene_bounds"):
        res.n_in_bounds
    # ... and a valid scene through the same pipeline afterwards is untouched
    ok = pipe.run(pipe.upload(synth_scene(H, H, seed=8)), (w / w.norm(dim=1, keepdim=True)).cuda(), seed=1)
    assert not bool(torch.isnan(ok.logits).any()) and ok.n_in_bounds > 0


def test_cu_partitioned_pipeline_is_bit_identical():
    """semabs_amd.scene.CuPartition: scene i's voxel stage on a stream masked to 64 CUs concurrently with scene i + 1's relevancy stage on the other 192 (the
    schedule of `bench.py --cu-split`, a measured negative result - profiles/r06_cu_split_ab.txt) gives the same result as the sequential schedule (relevancy maps
    and TSDF bit-identical, logits to the run-to-run spread of the GroupNorm atomics); the streams report their CU counts, which is what the persistent kernels
    size their grids with."""
    from semabs_amd.scene import CuPartition, build_default
    S, H, L, npts = 32, 96, 3, 4000
    pipe = build_default("ViT-B/32", precision="exact", chunk_tiles=64, max_labels=4, voxel=S, text_tower=False, num_input_pts=npts, config="ours")
    w = torch.randn(L, 512, generator=torch.Generator().manual_seed(0))
    w = (w / w.norm(dim=1, keepdim=True)).cuda()
    scenes = [pipe.upload(synth_scene(H, H, seed=20 + i)) for i in range(3)]
    seq = [pipe.run(s, w, seed=i) for i, s in enumerate(scenes)]
    torch.cuda.synchronize()
    part = CuPartition(64, "balanced")
    assert part.cus == (64, part.total - 64)
    cur = torch.cuda.current_stream()
    for s_ in part.streams:
        s_.wait_stream(cur)
    out, prev = [], None
    for i, s in enumerate(scenes):
        with torch.cuda.stream(part.vit):
            st = pipe.run_relevancy(s, w, seed=i)
        if prev is not None:
            with torch.cuda.stream(part.voxel):
                out.append(pipe.run_voxels(prev))
        prev = st
    with torch.cuda.stream(part.voxel):
        out.append(pipe.run_voxels(prev))
    for s_ in part.streams:
        cur.wait_stream(s_)
    torch.cuda.synchronize()
    for a, b in zip(seq, out):
        assert torch.equal(a.relevancies, b.relevancies) and torch.equal(a.tsdf, b.tsdf)
        # the UNet's GroupNorm statistics are fp64 sums accumulated with atomics: a few ulps of run-to-run spread on any schedule (conftest.bar's floor note)
        assert float((a.logits - b.logits).abs().max()) <= 2e-5 and float((a.labels == b.labels).float().mean()) > 0.9999
    part.close()


def test_uploaded_scene_with_a_changed_camera_does_not_use_stale_blocks():
    """ScenePipeline.upload caches the camera's device argument blocks; a caller that re-uses the dict with a new pose (a frame stream) must get the result of the
    NEW camera, i.e. the same as uploading the changed scene afresh."""
    from semabs_amd.scene import build_default
    S, H, L, npts = 32, 96, 2, 4000
    pipe = build_default("ViT-B/32", precision="exact", chunk_tiles=64, max_labels=4, voxel=S, text_tower=False, num_input_pts=npts, config="chefer_et_al")
    w = torch.randn(L, 512, generator=torch.Generator().manual_seed(0))
    w = (w / w.norm(dim=1, keepdim=True)).cuda()
    sc = synth_scene(H, H, seed=8)
    up = pipe.upload(sc)
    first = pipe.run(up, w, seed=1)
    pose = np.array(sc["cam_pose"], dtype=np.float64).copy()
    pose[:3, 3] += np.array([0.15, -0.1, 0.05])
    up["cam_pose"] = pose                                     # the caller mutates the uploaded dict
    moved = pipe.run(up, w, seed=1)
    fresh = pipe.run(pipe.upload(dict(sc, cam_pose=pose)), w, seed=1)
    assert torch.equal(moved.tsdf, fresh.tsdf) and float((moved.logits - fresh.logits).abs().max()) <= 2e-5
    assert not torch.equal(moved.tsdf, first.tsdf) and float((moved.logits - first.logits).abs().max()) > 1e-3
