"""Tuning aid (SEMABS_TUNE_LIB=1): level-0 convolution (k_conv16_lds) with its roles ablated: 0 full, 1 producers only, 2 consumers only,
4 consumers without epilogue.   python tools/conv16_probe.py [precision]"""
import os, sys
os.environ.setdefault("SEMABS_TUNE_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
from semabs_amd.unet3d import ResidualUNet3D
from semabs_amd.weights import make_semabs3d_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "exact"
P, S = 16, 128
u = ResidualUNet3D(16, 16, f_maps=16, num_groups=8, num_levels=6, precision=prec)
u.load_state_dict(make_semabs3d_state_dict(seed=3), prefix="vol_feature_extractor.")
u._sync()
conv = u.enc[0][1]
x = torch.randn(P, S, S, S, 16, device="cuda").to(u.act_dtype)
sums = torch.zeros(P, conv.groups, 2, dtype=torch.float64, device="cuda")
_lib.call("semabs_gn_stats", _lib.ptr(x), _lib.ptr(sums), P, S * S * S, 16, conv.groups, u.f32, _lib.stream())
for mode, label in [(0, "full"), (1, "producers only"), (2, "consumers only"), (4, "consumers, no epilogue"), (0, "full")]:
    _lib.call("semabs_conv_tune", 0, mode)
    for _ in range(2):
        y = u._conv(x, conv, relu=True, in_sums=sums)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = u._conv(x, conv, relu=True, in_sums=sums)
    e1.record(); torch.cuda.synchronize()
    print(f"{prec} {label:24s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us")
_lib.call("semabs_conv_tune", 0, 0)
