"""Tuning aid (SEMABS_TUNE_LIB=1 -> libsemabs_hip_tune.so): phase split of one k_conv_brick launch from s_memtime stamps of wave 0 of every
workgroup: {start, halo loads issued, halo in LDS, barrier passed, k-loop done, stores issued}.   python tools/conv_probe.py [level] [conv]"""
import os, sys
os.environ.setdefault("SEMABS_TUNE_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.unet3d import ResidualUNet3D
from semabs_amd.weights import make_semabs3d_state_dict

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ci = int(sys.argv[2]) if len(sys.argv) > 2 else 1
use_resid = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
persist = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # 1: the persistent-workgroup variant (tuning build only)
P, S = 16, 128 >> level
u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=6, precision="exact")
u.load_state_dict(make_semabs3d_state_dict(seed=3), prefix="vol_feature_extractor.")
u._sync()
conv = u.enc[level][ci]
_lib.call("semabs_conv_tune", 2, persist)
x = torch.randn(P, S, S, S, conv.cin, device="cuda")
resid = torch.randn(P, S, S, S, conv.cout, device="cuda") if use_resid else None
for _ in range(2):
    y = u._conv(x, conv, relu=True, resid=resid)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    y = u._conv(x, conv, relu=True, resid=resid)
e1.record(); torch.cuda.synchronize()
print(f"level {level} conv {ci} resid {int(use_resid)} persistent {persist}: {conv.cin} -> {conv.cout} at {S}^3 x {P}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch (incl. GroupNorm statistics)")
nwg = 1024 if persist else P * (S // 4) * (S // 8) * (S // 16) * max(1, conv.cout // 32)
tr = torch.zeros(nwg, 8, dtype=torch.int64, device="cuda")
scale, shift = u._gn(x, conv, None)                      # keep the GroupNorm statistics pass out of the traced launch's timing
sums = torch.zeros(P, conv.groups, 2, dtype=torch.float64, device="cuda")
_lib.call("semabs_gn_stats", _lib.ptr(x), _lib.ptr(sums), P, S * S * S, conv.cin, conv.groups, 1, _lib.stream())
_lib.call("semabs_conv_tune", 1, tr.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
y = u._conv(x, conv, relu=True, resid=resid, in_sums=sums)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
_lib.call("semabs_conv_tune", 1, 0)
t = tr.cpu().numpy().astype(np.float64)
t = t[t[:, 5] > 0]
span = t[:, 7].max() - t[:, 6].min()
life = (t[:, 7] - t[:, 6])
print(f"traced launch {us:.0f} us (conv + finalize); first start -> last end {span:.0f} ticks = {span / us:.0f} ticks / us; "
      f"workgroup lifetimes sum / (256 CUs x span) = {life.sum() / (256 * span):.2f} workgroups busy per CU on average")
d = np.diff(t[:, :6], axis=1)
names = ["(start ->) loads issued", "convert + ds_write", "barrier wait", "k-loop", "epilogue issue"]
print(f"{len(t)} workgroups; ticks of the last tile of each workgroup")
for i, n in enumerate(names):
    print(f"   {n:24s} mean {d[:, i].mean():9.0f}   median {np.median(d[:, i]):9.0f}")
print(f"   {'tile':24s} mean {(t[:, 5] - t[:, 0]).mean():9.0f}")
