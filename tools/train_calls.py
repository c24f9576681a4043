"""Per-call durations of the C-ABI launches of one VOOL training step (128^3, config 5), largest first:  python tools/train_calls.py [entry-substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.synth import SCENE_BOUNDS
from semabs_amd.train import VOOLTrainer
from semabs_amd.weights import make_semabsvool_state_dict
from train_bench import synth_batch

pat = sys.argv[1] if len(sys.argv) > 1 else ""
tr = VOOLTrainer(make_semabsvool_state_dict(seed=3), voxel_shape=(128,) * 3, scene_bounds=SCENE_BOUNDS)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(128, 80000, 400000, 4, seed=0).items()}
tr.step(batch); tr.step(batch); torch.cuda.synchronize()


class Hook:
    def __init__(self): self.rec = []
    def before(self, name, args):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        return name, tuple(a for a in args if isinstance(a, int) and abs(a) < (1 << 40))[:14], e0
    def after(self, tok):
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        self.rec.append(tok + (e1,))


h = Hook()
_lib.CALL_HOOK = h
tr.step(batch)
_lib.CALL_HOOK = None
torch.cuda.synchronize()
rows = sorted(((e0.elapsed_time(e1), name, ints) for name, ints, e0, e1 in h.rec if pat in name), reverse=True)
tot = {}
for ms, name, _ in rows:
    tot[name] = tot.get(name, 0.0) + ms
print("per entry point:", {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]})
for ms, name, ints in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{ms * 1e3:9.1f} us  {name:28s} {[i for i in ints if i < (1 << 31)]}")
