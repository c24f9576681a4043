"""How long does the host take to ISSUE one scene (no waiting for the GPU)?  python tools/cpu_issue_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd
from semabs_amd.scene import build_default
from semabs_amd.synth import synth_scene
pipe = build_default("ViT-B/16", precision="exact", chunk_tiles=220, max_labels=16, voxel=128, text_tower=False)
w = np.random.default_rng(0).standard_normal((16, 512)).astype(np.float32); w /= np.linalg.norm(w, axis=1, keepdims=True)
w_text = torch.from_numpy(w).cuda()
scenes = [pipe.upload(synth_scene(480, 480, seed=i)) for i in range(4)]
pipe.run(scenes[0], w_text, seed=0); torch.cuda.synchronize()
for i in range(1, 4):
    t0 = time.perf_counter()
    st = pipe.run_relevancy(scenes[i], w_text, seed=i)
    t1 = time.perf_counter()
    res = pipe.run_voxels(st)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"scene {i}: issue relevancy {1e3*(t1-t0):.1f} ms, issue voxels {1e3*(t2-t1):.1f} ms, then wait for the GPU {1e3*(t3-t2):.1f} ms, total {1e3*(t3-t0):.1f} ms")

# pure host cost: the same Python path with every C-ABI launch replaced by a no-op (allocations and torch ops still run)
from semabs_amd import _lib
real_call = _lib.call
n_calls = [0]
def fake(name, *a):
    n_calls[0] += 1
_lib.call = fake
import semabs_amd.clip.vit as _v, semabs_amd.clip as _c, semabs_amd.unet3d as _u, semabs_amd.net as _n, semabs_amd.scene as _s
for mod in (_v, _c, _u, _n, _s):
    if hasattr(mod, "_lib"):
        mod._lib.call = fake
torch.cuda.synchronize()
for i in range(1, 4):
    n_calls[0] = 0
    t0 = time.perf_counter()
    try:
        pipe.run(scenes[i], w_text, seed=i)
    except Exception as e:
        print("dry run stopped:", type(e).__name__, str(e)[:80])
    torch.cuda.synchronize()
    print(f"dry run {i}: host {1e3 * (time.perf_counter() - t0):.1f} ms for {n_calls[0]} C-ABI calls")
_lib.call = real_call
