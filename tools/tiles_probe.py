"""Time `semabs_tile_patches` alone on the headline tile table (480 x 480, "ours": 1 224 tiles x 2 flip passes, ViT-B/16 patch).
    python tools/tiles_probe.py [--patch 16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semabs_amd  # noqa: E402,F401
from semabs_amd import _lib  # noqa: E402
from semabs_amd.clip import ClipWrapper, plan_tiles, saliency_configs  # noqa: E402
from semabs_amd.synth import synth_jitter, synth_rgb  # noqa: E402
from semabs_amd.weights import make_clip_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--patch", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
H = W = 480
arch = "ViT-B/16" if a.patch == 16 else "ViT-B/32"
ClipWrapper.engine = None
ClipWrapper(arch, state_dict=make_clip_state_dict(arch, 0, text_tower=False), chunk_tiles=64, max_labels=4)
cfg = saliency_configs["ours"](H)
img = synth_rgb(H, W, seed=0)
imgs = np.stack([img] + [synth_jitter(img, k) for k in range(5)])
table, _ = plan_tiles(H, W, 6, cfg["cropping_augmentations"])
co = ClipWrapper._coeffs
ids = np.asarray([co.id_of(int(t)) for t in table[:, 3]], np.int32)
xmin_d, kk_d, ks_d = co.device()
tiles_dev = torch.from_numpy(np.concatenate([table, ids[:, None]], 1).astype(np.int32)).cuda()
g, p = 224 // a.patch, a.patch
n = len(table)
imgs_d = torch.from_numpy(imgs).cuda()
patches = torch.zeros(2 * n * g * g, 3 * p * p, dtype=torch.float16, device="cuda")


def run():
    _lib.call("semabs_tile_patches", _lib.ptr(imgs_d), 6, H, W, _lib.ptr(tiles_dev), n, _lib.ptr(xmin_d), _lib.ptr(kk_d), _lib.ptr(ks_d),
              _lib.ptr(ClipWrapper._lut), _lib.ptr(patches), p, 2, max(co.ksize), _lib.stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / a.iters * 1e3
gb = patches.numel() * 2 / 1e9
print(f"tile_patches: {n} tiles x 2 passes, {us:.1f} us per launch, {gb:.2f} GB written = {gb / us * 1e3:.2f} TB/s; checksum {float(patches.float().sum()):.6e}")
