"""The same audit as tools/aten_audit.py for ONE steady-state VOOL training step (128^3, config 5): every ATen operator / memcpy / memset it issues, with
the innermost call site inside this repository.   python tools/aten_audit_train.py [out.txt]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from torch.profiler import ProfilerActivity, profile
import semabs_amd  # noqa
from semabs_amd.synth import SCENE_BOUNDS
from semabs_amd.train import VOOLTrainer
from semabs_amd.weights import make_semabsvool_state_dict
from train_bench import synth_batch

tr = VOOLTrainer(make_semabsvool_state_dict(seed=3), voxel_shape=(128,) * 3, scene_bounds=SCENE_BOUNDS)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(128, 80000, 400000, 4, seed=0).items()}
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
sites, kernels, ktime = collections.Counter(), collections.Counter(), collections.Counter()
for ev in prof.events():
    name = ev.name
    if ev.device_type == torch.autograd.DeviceType.CUDA or name.startswith(("Memcpy", "Memset")):
        if not (name.startswith(("k_", "void k_", "_Z")) or "semabs" in name):
            kernels[name[:110]] += 1
            ktime[name[:110]] += ev.device_time if hasattr(ev, "device_time") else 0
        continue
    if not name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    site = next((s for s in (ev.stack or []) if "semantic-abstraction_amd" in s or "train_bench" in s or "aten_audit" in s), "?")
    sites[(name, site.strip()[-110:], str(ev.input_shapes)[:60])] += 1
out = ["non-product device activities of one training step (ATen kernels, memcpy, memset; count, total us):"]
out += [f"  {n:4d}  {ktime[k]:9.1f}  {k}" for k, n in kernels.most_common()]
out += ["", "top-level ATen operators by call site:"]
out += [f"  {n:4d}  {k[0]:28s} {k[2]:60s} {k[1]}" for k, n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0]))]
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
