"""Scratch timing of the relevancy stage at the BASELINE shape (used while tuning; bench.py is the judged entry)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd
from semabs_amd.clip import ClipWrapper, saliency_configs
from semabs_amd.weights import make_clip_state_dict
from semabs_amd.synth import synth_rgb

arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/16"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sd = make_clip_state_dict(arch, 0, text_tower=False)
ClipWrapper(arch, state_dict=sd, chunk_tiles=chunk, max_labels=16)
ClipWrapper.n_streams = int(os.environ.get("SEMABS_STREAMS", "2"))
L = 16
w = torch.randn(L, 512, device="cuda"); w = w / w.norm(dim=-1, keepdim=True)
img = synth_rgb(480, 480, 1)
cfg = saliency_configs["ours"](480)
images = ClipWrapper.make_images(img, cfg["augmentations"])
for r in range(reps + 1):
    torch.cuda.synchronize(); t = time.time()
    maps = ClipWrapper.relevancy_device(images, w, cfg["cropping_augmentations"], True, True)
    torch.cuda.synchronize(); dt = time.time() - t
    fl = 2448 * (35.127e9 if arch.endswith("16") else 8.818e9)
    print(f"{arch} chunk={chunk} streams={ClipWrapper.n_streams} run{r}: {dt*1e3:.1f} ms  -> {1/dt:.2f} scenes/s, {fl/dt/1e12:.1f} TFLOP/s (algorithmic)", flush=True)
print("maps", tuple(maps.shape), float(maps.abs().max()))
