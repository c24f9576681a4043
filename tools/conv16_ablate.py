"""Level-0 convolution (k_conv16_lds, exact mode) role ablation: python tools/conv16_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.unet3d import ResidualUNet3D
from semabs_amd.weights import make_semabs3d_state_dict
u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=6, precision="exact")
u.load_state_dict(make_semabs3d_state_dict(seed=3), prefix="vol_feature_extractor.")
P, S = 16, 128
x = torch.randn(P, S, S, S, 16, device="cuda")
conv = u.enc[0][1]
scale, shift = u._gn(x, conv)
y = torch.empty_like(x)
def run():
    _lib.call("semabs_conv3d", _lib.ptr(x), _lib.ptr(conv.w_hi), _lib.ptr(conv.w_lo), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift),
              None, None, P, S, S, S, 16, 16, 3, 1, 1, _lib.stream())
for ab, label in [(0, "full"), (1, "producers only (no consume)"), (2, "consumers only (no produce)"), (4, "consumers only, no epilogue"), (0, "full")]:
    _lib.call("semabs_conv_set_config", 1 + 2 * ab)
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:36s} {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
_lib.call("semabs_conv_set_config", 1)
r = torch.randn_like(x)
def run_res():
    _lib.call("semabs_conv3d", _lib.ptr(x), _lib.ptr(conv.w_hi), _lib.ptr(conv.w_lo), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift),
              None, _lib.ptr(r), P, S, S, S, 16, 16, 3, 1, 1, _lib.stream())
for _ in range(2): run_res()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run_res()
e1.record(); torch.cuda.synchronize()
print(f"{'full, with residual':36s} {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
sums = torch.zeros(P, 8, 2, dtype=torch.float64, device="cuda")
def run_stats():
    _lib.call("semabs_conv3d_stats", _lib.ptr(x), _lib.ptr(conv.w_hi), _lib.ptr(conv.w_lo), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift),
              None, None, P, S, S, S, 16, 16, 3, 1, 1, _lib.ptr(sums), 8, _lib.stream())
for fn, label in ((run, "full"), (run_stats, "full + fused output statistics"), (run, "full"), (run_stats, "full + fused output statistics")):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:36s} {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
