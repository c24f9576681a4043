"""One scene split over two ranks (tile-sharded relevancy + label-sharded voxel inference) equals the single-rank result.
Two processes share the one GPU of the test box, so the process group is gloo (RCCL refuses two ranks on one device); the code path is the
one `nccl` takes on a multi-GPU node: ClipWrapper.relevancy_device(tile_range=...) -> all-gather -> aggregate, per-rank label slices ->
all-gather."""
import os

import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401

pytestmark = pytest.mark.gpu
IMG, VOX, L = 96, 32, 3


def _pipeline(config="ours"):
    from semabs_amd.scene import build_default
    return build_default("ViT-B/32", precision="exact", chunk_tiles=64, max_labels=4, voxel=VOX, text_tower=False, num_input_pts=4000, config=config)


def _inputs():
    from semabs_amd.synth import synth_scene
    rng = np.random.default_rng(0)
    w = rng.standard_normal((L, 512)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    return synth_scene(IMG, IMG, seed=11), torch.from_numpy(w).cuda()


def _worker(rank, world, port, q, config="ours"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = _pipeline(config)
        scene, w = _inputs()
        res = pipe.run_sharded(pipe.upload(scene), w, seed=5)
        torch.cuda.synchronize()
        q.put((rank, res.relevancies.cpu().numpy(), res.logits.cpu().numpy(), res.labels.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,config", [(2, "ours"), (3, "chefer_et_al"), (8, "ours")])
def test_sharded_scene_equals_single_rank(world, config):
    """world 2: the plain case.  world 3 with the single-scale config: ONE tile for three ranks -> two ranks run no ViT forward at all (empty
    tile shard), and 3 labels / 3 ranks.  world 8: 1 224 tiles / 8 ranks, but only 3 label volumes -> five ranks with an EMPTY label shard
    contribute zero rows to the logit all-gather (scene.py run_voxels, dist.shard_range)."""
    import torch.multiprocessing as mp
    pipe = _pipeline(config)
    scene, w = _inputs()
    ref = pipe.run(pipe.upload(scene), w, seed=5)
    ref_maps, ref_logits, ref_labels = ref.relevancies.cpu().numpy(), ref.logits.cpu().numpy(), ref.labels.cpu().numpy()
    del pipe
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, config)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for rank, maps, logits, labels in got:
        # a tile's relevance does not depend on the other tiles of its batch, and the all-gather only moves it; what differs is which GEMM
        # kernel a (smaller) batch selects, i.e. the fp32 summation order inside the MFMAs, seen through the fp16 canvases of the
        # aggregation (measured 1.2e-4 of the maximum = 6e-7 absolute; the bar of the path is 1e-3 absolute)
        assert np.abs(maps - ref_maps).max() <= 5e-4 * np.abs(ref_maps).max(), (rank, np.abs(maps - ref_maps).max(), np.abs(ref_maps).max())
        # the label volumes are independent; GroupNorm statistics are reduced with floating-point atomics -> equal to rounding
        np.testing.assert_allclose(logits, ref_logits, rtol=0, atol=2e-4 * max(1.0, float(np.abs(ref_logits).max())))
        assert (labels != ref_labels).mean() < 2e-3
    assert all(np.array_equal(got[0][2], g_[2]) for g_ in got[1:])      # every rank holds the same gathered logits
