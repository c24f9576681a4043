"""Hot-path hygiene audit: every ATen operator / memcpy / memset issued by ONE steady-state scene of the benchmark (text tower +
ScenePipeline.run), with the innermost call site inside this repository.  Target: none.   python tools/aten_audit.py [out.txt]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile
import semabs_amd  # noqa
from semabs_amd.scene import build_default
from semabs_amd.synth import synth_scene

pipe = build_default("ViT-B/16", precision="exact", chunk_tiles=2448, max_labels=16, voxel=128, text_tower=False)
w = np.random.default_rng(0).standard_normal((16, 512)).astype(np.float32)
w_text = torch.from_numpy(w / np.linalg.norm(w, axis=1, keepdims=True)).cuda()
scenes = [pipe.upload(synth_scene(480, 480, seed=i)) for i in range(3)]
text_enc = tokens = None
tk = os.path.join(ROOT, "tests", "golden", "tokens_default.npz")
if os.path.exists(tk):
    from semabs_amd.clip.vit import TextEncoder
    from semabs_amd.weights import make_clip_state_dict
    text_enc = TextEncoder(make_clip_state_dict("ViT-B/16", 0, text_tower=True))
    tokens = torch.from_numpy(np.load(tk)["tokens"][:16])


def step(i):
    wt = text_enc.zeroshot_weights(tokens, 16, 1) if text_enc is not None else w_text
    return pipe.run(scenes[i], wt, seed=i)


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()
sites = collections.Counter()
kernels = collections.Counter()
for ev in prof.events():
    name = ev.name
    if ev.device_type == torch.autograd.DeviceType.CUDA or name.startswith(("Memcpy", "Memset")):
        if not (name.startswith(("k_", "void k_", "_Z")) or "semabs" in name):
            kernels[name[:110]] += 1
        continue
    if not name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    site = next((s for s in (ev.stack or []) if "semantic-abstraction_amd" in s or "bench.py" in s or "aten_audit" in s), "?")
    sites[(name, site.strip()[-110:], str(ev.input_shapes)[:60])] += 1
out = ["non-product device activities of one scene (ATen kernels, memcpy, memset):"]
out += [f"  {n:4d}  {k}" for k, n in kernels.most_common()]
out += ["", "top-level ATen operators by call site:"]
out += [f"  {n:4d}  {k[0]:28s} {k[2]:60s} {k[1]}" for k, n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0]))]
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
