#!/usr/bin/env python
"""bench.py — scenes/s of the relevancy -> fusion -> OVSSC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE synthetic scene end to end on one GPU: 480x480 RGB-D, 16 labels, ViT-B/16, the "ours" saliency config
(204 tiles x 6 images (5 colour-jitter copies) x 2 flips = 2 448 ViT forwards), analytic attention x gradient rollout,
multi-scale aggregation, depth -> points -> voxel indices, point MLP, scatter-mean, 6-level ResidualUNet3D on 16 label
volumes of 128^3, implicit decoder at the 128^3 voxel centres, TSDF integration and the OVSSC post-mask.  Inputs
(uint8 images, fp32 depth, weights) are resident in HBM before the timed region.  Ranks process disjoint scenes
(scene sharding, no data-path collective) -> weak scaling; value = all scenes / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the fp16 MFMA GEMM, timed
live inside the timed region with HIP events carried by the launches' own dispatch packets) and `cpu_baseline` (the oracle on the host cores, bounded
sample, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import semabs_amd  # noqa: E402,F401

ARCH = "ViT-B/16"
N_LABELS = 16
IMG = 480
VOXEL = 128
FLOPS_PER_TILE = {"ViT-B/16": 35.127e9, "ViT-B/32": 8.818e9}      # SURVEY.md §8d per-tile forward count
PEAK_F16_TFLOPS = 2500.0                                          # dense MFMA peak (MI355X_MICROARCH.md)


def cpu_baseline(arch, n_labels, tiles_sample=16):
    """Oracle ("port" of the reference's CPU path) timed on this host: bounded sample, scaled to one scene."""
    from oracle import relevancy as orl
    from oracle import semabs3d as os3
    from semabs_amd.synth import synth_rgb
    from semabs_amd.weights import make_clip_state_dict, make_semabs3d_state_dict
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    sd = make_clip_state_dict(arch, 0, text_tower=False)
    nsd = make_semabs3d_state_dict(seed=3)
    rng = np.random.default_rng(0)
    w = torch.from_numpy(rng.standard_normal((512, n_labels)).astype(np.float32))
    img = synth_rgb(IMG, IMG, 1)
    cfg = orl.saliency_configs["ours"](IMG)
    table = orl.tile_table(IMG, IMG, 1, cfg["cropping_augmentations"])
    pick = table[np.linspace(0, len(table) - 1, tiles_sample).astype(int)]
    t0 = time.time()
    tiles = orl.make_tile_images([img], pick)
    t_pre = time.time() - t0
    with torch.no_grad():
        orl.gradcam_tiles(sd, tiles[:2], w, True)                                   # warm-up
        t0 = time.time()
        orl.gradcam_tiles(sd, tiles, w, True)
        t_vit = time.time() - t0
        x = torch.zeros(1, 16, VOXEL, VOXEL, VOXEL)
        x[0, :, ::5, ::7, ::3] = 1.0
        t0 = time.time()
        os3.unet_forward(nsd, x, 6)
        t_unet = time.time() - t0
    n_fwd = 2448
    scene_s = (t_pre / tiles_sample) * 1224 + (t_vit / tiles_sample) * n_fwd + t_unet * n_labels
    return {"value": 1.0 / scene_s, "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": f"{tiles_sample} of 2448 tile forwards ({arch}, {n_labels} labels, analytic rollout) + 1 of {n_labels} "
                      f"128^3 UNet volumes, torch-CPU fp32 oracle, scaled to one scene "
                      f"(pre {t_pre:.2f}s, vit {t_vit:.2f}s, unet {t_unet:.2f}s); aggregation/decoder not included",
            "scene_seconds_estimate": scene_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arch", default=ARCH)
    ap.add_argument("--precision", default="exact", choices=["exact", "fp16"], help="UNet arithmetic (see DESIGN.md)")
    ap.add_argument("--chunk", type=int, default=2448, help="tile forwards per ViT batch.  2448 = the whole scene in one batch (11 GB of activations of "
                    "the 288 GB): the GEMMs run 22 / 66 / 88 waves of tiles deep, so launch, prologue and last-wave effects are amortised - "
                    "901 vs 853 TFLOP/s and 136.7 vs 142.7 ms per scene against 220 (which was tuned to have no ragged last wave: 1.99 / "
                    "5.98 / 7.97 waves); 663 / 1224: 139 ms.  The maps are bit-identical for every chunk size (tools/chunk_equiv.py)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the tile chunks are pipelined over when --chunk cuts the scene into several "
                    "batches (2 = +4%% scenes/s at --chunk 220, but overlapping kernels blur the per-launch timing of the roofline leg)")
    ap.add_argument("--time-every", type=int, default=1, help="roofline leg: attach timing events to one GEMM launch in n of the timed region (chosen by a "
                    "hash of the launch index).  1 = every launch: free with one ViT batch per scene (~60 launches); with small batches "
                    "(--chunk 220: 660 launches per scene) it costs 1.9 %% of scenes/s - every dispatch packet then carries a completion signal - "
                    "and 5 gives the same TFLOP/s figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback")
    if args.gpus > 1 and "RANK" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL), exactly as the driver would
        # (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...), and hand back its exit code.
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run plainly and let bench.py spawn the ranks)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from semabs_amd.clip import vit as vitmod
    from semabs_amd.scene import build_default
    from semabs_amd.synth import synth_scene
    pipe = build_default(args.arch, precision=args.precision, chunk_tiles=args.chunk, max_labels=N_LABELS, voxel=VOXEL, text_tower=False)
    rng = np.random.default_rng(0)
    w = rng.standard_normal((N_LABELS, 512)).astype(np.float32)          # synthetic unit-norm zero-shot text weights
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    w_text = torch.from_numpy(w).cuda()
    n_scenes = args.steps + args.warmup
    from semabs_amd.clip import ClipWrapper, saliency_configs
    ClipWrapper.n_streams = max(1, args.streams)
    cfg = saliency_configs["ours"](IMG)
    scenes = [pipe.upload(synth_scene(IMG, IMG, seed=1000 * rank + i)) for i in range(n_scenes)]      # RGB-D frames resident in HBM

    def step(i):                                        # everything from the raw frame on is inside the timed region (incl. the colour jitter)
        return pipe.run(scenes[i], w_text, seed=i)

    def run_range(lo, hi):
        res = None
        for i in range(lo, hi):
            res = step(i)
        return res

    run_range(0, args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    timer = vitmod.GemmTimer(every=args.time_every)
    vitmod.GEMM_TIMER = timer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_range(args.warmup, n_scenes)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    vitmod.GEMM_TIMER = None
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gs = timer.summary()
    total_scenes = args.steps * world
    value = total_scenes / dt
    if rank == 0:
        ach = gs["flops"] / (gs["total_ms"] * 1e-3) / 1e12 if gs["total_ms"] > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "gemm_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "scenes/sec (relevancy+3D-UNet infer), 480x480x16-label x128^3",
            "value": value, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f16 (MFMA operands, fp32 accumulate; UNet hi/lo-split = fp32-equivalent)",
            "data": "synthetic",
            "config": {"workload": f"end-to-end relevancy->fusion->OVSSC per scene: {IMG}x{IMG} RGB-D, {N_LABELS} labels, {args.arch}, "
                                   f"'ours' saliency config (2448 tile forwards), {VOXEL}^3 voxels, 80000 input points; scene-sharded",
                       "arch": args.arch, "unet_precision": args.precision, "tile_chunk_streams": args.streams, "scenes_per_gpu": args.steps, "parallelism": f"scene-shard x{world}"},
            "relevancy_tflops_algorithmic": 2448 * FLOPS_PER_TILE[args.arch] * total_scenes / dt / 1e12,
            "roofline": {"kernel": "fp16 GEMM: k_gemm8 (large shapes) + k_gemm_f16 (small), all epilogues", "bound": "mfma", "achieved": ach, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_F16_TFLOPS, "traffic": traffic, "launches": gs["launches"], "launches_in_timed_region": gs["seen"],
                         "sampling": ("every GEMM launch of the timed region carries start / stop events" if args.time_every == 1 else
                                      f"1 in {args.time_every} GEMM launches of the timed region (hashed launch index) carries start / stop events"),
                         "avg_launch_us": gs["total_ms"] * 1e3 / max(1, gs["launches"]),
                         "gemm_share_of_step": gs["total_ms"] * 1e-3 * gs["seen"] / max(1, gs["launches"]) / dt if world == 1 else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.arch, N_LABELS)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
