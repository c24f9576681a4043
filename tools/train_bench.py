"""Time the VOOL training step (config 4 of BASELINE.json: train_vool.py defaults - 128^3 voxels, batch 1, 4 descriptions, 80 000 input
points, 400 000 query points, UNet f_maps 16 x 6 levels, LAMB lr 1e-3) on one MI355X with synthetic data.

    python tools/train_bench.py [--steps 3] [--warmup 1] [--S 128] [--N 80000] [--M 400000] [--D 4]
Data parallel (config 4 asks for 8 GPUs): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
    --master-port P tools/train_bench.py ...   - one rank per GPU, a different synthetic batch per rank, ONE RCCL all-reduce of the flat
gradient buffer per step (semabs_amd.dist.allreduce_flat_gradients); rank 0 prints the line (max-over-ranks step time, all-reduce time).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semabs_amd  # noqa: E402,F401
from semabs_amd.synth import SCENE_BOUNDS  # noqa: E402
from semabs_amd.train import RELATIONS, VOOLTrainer  # noqa: E402
from semabs_amd.weights import make_semabsvool_state_dict  # noqa: E402


def synth_batch(S, N, M, D, seed=0):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(SCENE_BOUNDS[0]), np.array(SCENE_BOUNDS[1])
    # points on a few random planes (a depth-camera-like surface sample), not a uniform cloud: ~3 % of the voxels occupied
    base = rng.random((N, 3))
    base[np.arange(N), rng.integers(0, 3, N)] *= 0.05      # one coordinate PER POINT (until round 3 this line scaled every column: 80 000 points in a 6^3-voxel cube)
    xyz = (lo + (hi - lo) * np.clip(base, 0, 1)).astype(np.float32)[None]
    sal = rng.random((1, 2 * D, N, 1)).astype(np.float32)
    q = (lo + (hi - lo) * rng.random((1, D, M, 3))).astype(np.float32)
    label = (rng.random((1, D, M)) < 0.2).astype(np.float32)
    rels = [[RELATIONS[i % 6]] for i in range(D)]
    return dict(input_xyz_pts=torch.from_numpy(xyz), input_target_saliency_pts=torch.from_numpy(sal[:, :D]),
                input_reference_saliency_pts=torch.from_numpy(sal[:, D:]), output_xyz_pts=torch.from_numpy(q),
                output_label_pts=torch.from_numpy(label), spatial_relation_name=rels)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--S", type=int, default=128)
    ap.add_argument("--N", type=int, default=80000)
    ap.add_argument("--M", type=int, default=400000)
    ap.add_argument("--D", type=int, default=4)
    a = ap.parse_args()
    import os
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    tr = VOOLTrainer(make_semabsvool_state_dict(seed=3), voxel_shape=(a.S,) * 3, scene_bounds=SCENE_BOUNDS)
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(a.S, a.N, a.M, a.D, seed=rank).items()}
    losses = []
    for _ in range(a.warmup):
        losses.append(float(tr.step(batch)["loss"]))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    fb = 0.0
    for _ in range(a.steps):
        t1 = time.perf_counter()
        out = tr.forward_backward(batch)
        torch.cuda.synchronize()
        fb += time.perf_counter() - t1
        out["gradnorm"] = tr.optimizer_step()                   # all-reduce (world > 1) + clip + LAMB
        losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / a.steps, fb / a.steps], device="cuda")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        step_ms, fb_ms = float(dt[0]) * 1e3, float(dt[1]) * 1e3
        print(json.dumps({"metric": "vool_train_step_ms", "value": step_ms, "fwd_bwd_ms": fb_ms, "allreduce_clip_lamb_ms": step_ms - fb_ms,
                          "steps_per_s_all_ranks": world / (step_ms / 1e3), "n_gpus": world, "steps": a.steps,
                          "config": {"S": a.S, "N": a.N, "M": a.M, "D": a.D}, "losses": [round(x, 5) for x in losses],
                          "gradnorm": float(out["gradnorm"]), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))
    if world > 1:
        dist.destroy_process_group()
