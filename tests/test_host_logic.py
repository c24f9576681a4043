"""CPU: host-side logic of the product (no GPU): tile planning vs the oracle / golden tables, tokenizer, sharding,
weight generators, UNet layer plan, and a world_size-2 gloo run of the sharding collectives."""
import os
import sys

import numpy as np
import pytest
import torch

import semabs_amd  # noqa: F401
from oracle import relevancy as orl


@pytest.mark.parametrize("H,W,cfgname,dim,n_img", [(480, 480, "ours", 480, 6), (120, 120, "ours", 120, 2), (256, 192, "ours", 256, 2),
                                                   (96, 96, "chefer_et_al", 96, 1), (100, 130, "ours", 100, 2), (64, 64, "ours", 64, 1)])
def test_plan_tiles_matches_oracle(H, W, cfgname, dim, n_img):
    from semabs_amd.clip import plan_tiles, saliency_configs
    cfg = saliency_configs[cfgname](dim)
    assert cfg == orl.saliency_configs[cfgname](dim)
    table, scales = plan_tiles(H, W, n_img, cfg["cropping_augmentations"])
    assert np.array_equal(table, orl.tile_table(H, W, n_img, cfg["cropping_augmentations"]))
    per_img = len(table) // n_img
    assert sum(int(s[2]) * int(s[3]) for s in scales) == per_img
    # scale descriptors reproduce the table: tile = base + col_idx * n_rows + row_idx
    for ts, stride, nx, ny, base in scales:
        for iy in range(ny):
            for ix in range(nx):
                im, x, y, t = table[base + iy * nx + ix]
                assert (im, x, y, t) == (0, ix * stride, iy * stride, ts)


def test_plan_tiles_golden(golden):
    from semabs_amd.clip import plan_tiles, saliency_configs
    g = golden("g1_tiling")
    table, _ = plan_tiles(480, 480, 2, saliency_configs["ours"](480)["cropping_augmentations"])
    assert np.array_equal(table, g["table_480"])


def test_duplicate_tile_size_rejected():
    from semabs_amd.clip import plan_tiles
    with pytest.raises(NotImplementedError):
        plan_tiles(64, 64, 1, [{"tile_size": 32, "stride": 8}, {"tile_size": 32, "stride": 16}])


def test_tokenizer_against_golden(golden):
    from semabs_amd.clip.tokenizer import BPETokenizer, find_vocab
    if find_vocab() is None:
        pytest.skip("CLIP BPE merge table not present")
    g = golden("g7_text")
    tk = BPETokenizer()
    labels = ["chair", "table", "pink make up bag", "brown modern upholstered chair in faux leather with wooden legs"]
    P = "a photograph of a {} in a home."
    assert np.array_equal(tk.tokenize([P.format(c) for c in labels]).numpy(), g["t1_tokens"])
    t3 = tk.tokenize([t.format(c) for c in labels for t in ["a photo of a {}.", "a bad photo of the {}.", P]]).numpy()
    assert np.array_equal(t3, g["t3_tokens"])
    assert np.array_equal(tk.tokenize(["Hello, World! it's 42 degrees", "a  b\tc", "don't you're we've"]).numpy(), g["misc_tokens"])
    with pytest.raises(RuntimeError):
        tk.tokenize("word " * 100)
    assert tk.tokenize("word " * 100, truncate=True).shape == (1, 77)


def test_imagenet_templates_table_and_reference_import_line(golden):
    """`from CLIP.clip import ClipWrapper, saliency_configs, imagenet_templates` (generate_relevancy.py:10) must keep working after the swap;
    the 80 templates and their order are the interface (the zero-shot weight is their mean): digest pinned to the reference's table (g24)."""
    import hashlib
    from semabs_amd.clip import ClipWrapper, imagenet_templates, saliency_configs  # noqa: F401
    import semabs_amd.clip as pkg
    g = golden("g24_prompt_ensemble")
    assert len(imagenet_templates) == int(g["n_templates"]) == 80 and "imagenet_templates" in pkg.__all__
    assert np.array_equal(np.frombuffer(hashlib.sha256("\n".join(imagenet_templates).encode()).digest(), dtype=np.uint8), g["templates_sha"])
    assert all(t.count("{}") == 1 for t in imagenet_templates)
    assert all(cfg(480)["imagenet_prompt_ensemble"] is False for cfg in saliency_configs.values())


def test_tokenizer_on_the_prompt_ensemble(golden):
    from semabs_amd.clip import imagenet_templates
    from semabs_amd.clip.tokenizer import BPETokenizer, find_vocab
    if find_vocab() is None:
        pytest.skip("CLIP BPE merge table not present")
    g = golden("g24_prompt_ensemble")
    texts = [t.format(c) for c in [str(l) for l in g["labels"]] for t in imagenet_templates]
    assert np.array_equal(BPETokenizer().tokenize(texts).numpy(), g["tokens"])


def test_shard_range_partitions():
    from semabs_amd.dist import shard_list, shard_range
    for n in (0, 1, 7, 8, 64, 2448):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    assert shard_list(list(range(10)), 1, 3) == [4, 5, 6]


def test_weight_generators_are_deterministic_and_complete():
    from semabs_amd.unet3d import ResidualUNet3D
    from semabs_amd.weights import make_clip_state_dict, make_semabs3d_state_dict, unet_layer_plan
    a = make_clip_state_dict("ViT-B/32", 0, text_tower=False)
    b = make_clip_state_dict("ViT-B/32", 0, text_tower=False)
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["visual.transformer.resblocks.3.mlp.c_fc.weight"]
    assert torch.equal(w, w.half().float())                                  # fp16-representable like convert_weights leaves them
    plan = unet_layer_plan(16, 16, 16, 6)
    assert sum(1 for p in plan if p[1] in ("gcr", "gc")) == 33 and sum(1 for p in plan if p[1] == "convT") == 5
    sd = make_semabs3d_state_dict(seed=3)
    n_params = sum(v.numel() for k, v in sd.items() if k != "steps")
    assert n_params == 35403969                                              # SemAbs3D parameter count
    u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=6)
    assert set("vol_feature_extractor." + k for k in u.expected_keys()) <= set(sd)


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import semabs_amd  # noqa: F401
    from semabs_amd.dist import allgather_tile_relevance, allreduce_flat_gradients, gather_results, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, N, g = 3, 11, 2                                                        # 11 tiles: ranks hold 6 and 5 (padded slices)
    full = torch.arange(L * N * g * g, dtype=torch.float32).view(L, N, g, g)
    lo, hi = shard_range(N, rank, world)
    mine = full[:, lo:hi].contiguous()                                        # this rank's tile slice (10 tiles over 2 ranks; 3 ranks: ragged)
    rel = allgather_tile_relevance([mine, 2 * mine], N)                       # two passes (flip) travel in one collective
    ok1 = torch.equal(rel[0], full) and torch.equal(rel[1], 2 * full)
    gathered = gather_results(torch.full((2, 4), float(rank)))               # e.g. 2 label volumes per rank
    ok2 = gathered.shape == (world, 2, 4) and all(float(gathered[r, 0, 0]) == r for r in range(world))
    # training: one flat all-reduce of gradients + usage flags (rank 0 used relation 0, rank 1 relation 2; relation 1 unused everywhere)
    flat = torch.cat([torch.arange(6, dtype=torch.float32) * (rank + 1), torch.tensor([1.0, 0.0, 0.0] if rank == 0 else [0.0, 0.0, 1.0])])
    scale, used = allreduce_flat_gradients(flat, 3)
    ok3 = scale == 0.5 and torch.equal(flat[:6] * scale, torch.arange(6, dtype=torch.float32) * 1.5) and used.tolist() == [True, False, True]
    # the same exchange in buckets announced in backward-completion order (tail of the buffer first, head last) and started asynchronously where the
    # backend can: bit-identical to the single call, for any announcement pattern (none / some / all before finish); per-process accounting filled
    from semabs_amd import dist as sdist
    from semabs_amd.dist import BucketedAllReduce
    gen = torch.Generator().manual_seed(7 + rank)
    base = torch.randn(1000, generator=gen)
    want = base.clone()
    allreduce_flat_gradients(want, 0)
    ok4 = True
    for announce in ((), (0, 2), (0, 1, 2, 3), (3, 1)):
        buf = base.clone()
        br = BucketedAllReduce(buf, [(600, 1000), (250, 600), (100, 250), (0, 100)])
        br.begin_step()
        for i in announce:
            br.ready(i)
        ok4 = ok4 and br.finish() == 0.5 and torch.equal(buf, want)
    st = sdist.stats_snapshot()
    ok4 = ok4 and st["all_reduce_bucket"]["bytes"] == 4 * 4000 and st["all_reduce"]["calls"] >= 2 and st["all_gather"]["bytes"] > 0
    q.put((rank, bool(ok1), bool(ok2 and ok3 and ok4)))
    dist.destroy_process_group()


def test_sharding_collectives_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True, True), (1, True, True)]


def test_pack_fragments_layout():
    """unet3d._pack_fragments: [Cout, Kp] -> [Kp / 32][Cout / 16][lane = kg * 16 + row][8] with k = 32 * k-step + 8 * kg + e - the order in which
    a wave holds the A operand of v_mfma_f32_16x16x32_f16 (SEMABS_CONV_PACKED, include/semabs.h)."""
    import torch
    from semabs_amd.unet3d import _pack_fragments
    cout, kp = 48, 96
    w = torch.arange(cout * kp, dtype=torch.float32).view(cout, kp)
    p = _pack_fragments(w).view(kp // 32, cout // 16, 64, 8)
    for ks in range(kp // 32):
        for cb in range(cout // 16):
            for lane in (0, 5, 16, 37, 63):
                row, kg = lane & 15, lane >> 4
                for e in (0, 3, 7):
                    assert p[ks, cb, lane, e] == w[cb * 16 + row, ks * 32 + kg * 8 + e]
    assert p.numel() == w.numel()
