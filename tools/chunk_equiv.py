"""The relevancy maps must not depend on the ViT batch (chunk) size: python tools/chunk_equiv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd  # noqa
from semabs_amd.scene import build_default
from semabs_amd.synth import synth_scene
rng = np.random.default_rng(0)
w = rng.standard_normal((16, 512)).astype(np.float32); w /= np.linalg.norm(w, axis=1, keepdims=True)
outs = []
for chunk in (220, 2448):
    pipe = build_default("ViT-B/16", precision="exact", chunk_tiles=chunk, max_labels=16, voxel=128, text_tower=False)
    res = pipe.run(pipe.upload(synth_scene(480, 480, seed=3)), torch.from_numpy(w).cuda(), seed=3)
    torch.cuda.synchronize()
    outs.append((res.relevancies.cpu().numpy().copy(), res.logits.cpu().numpy().copy()))
    print(chunk, "peak GB", torch.cuda.max_memory_allocated() / 1e9, flush=True)
    del pipe, res
    torch.cuda.empty_cache()
m0, l0 = outs[0]; m1, l1 = outs[1]
print("maps bit-identical:", np.array_equal(m0, m1), " max |diff|", float(np.abs(m0 - m1).max()), " max |map|", float(np.abs(m0).max()))
print("logits max |diff|", float(np.abs(l0 - l1).max()), " (GroupNorm statistics use floating-point atomics)")
