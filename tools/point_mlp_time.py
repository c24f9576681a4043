"""Point MLP (4 -> 128 -> 128 -> 16) at the bench shape, MFMA kernel vs fp32 FMA kernel: python tools/point_mlp_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd  # noqa
from semabs_amd import _lib
P, N = 16, 80000
rng = np.random.default_rng(0)
t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
xyz, feat = t(N, 3), t(P, N)
w1, b1, w2, b2, w3, b3 = t(128, 4) * 0.5, t(128) * 0.1, t(128, 128) * 0.09, t(128) * 0.1, t(16, 128) * 0.09, t(16) * 0.1
outs = []
for cfg, label in ((1 + 256, "fp32 FMA"), (1, "MFMA hi/lo")):
    _lib.call("semabs_conv_set_config", cfg)
    out = torch.empty(P, N, 16, device="cuda")
    def run():
        _lib.call("semabs_point_mlp", _lib.ptr(xyz), _lib.ptr(feat), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(w3), _lib.ptr(b3),
                  _lib.ptr(out), P, N, 128, 16, _lib.stream())
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:12s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
    outs.append(out.double().cpu())
_lib.call("semabs_conv_set_config", 1)
h = torch.nn.functional.leaky_relu(torch.cat([xyz.double().cpu()[None].expand(P, N, 3), feat.double().cpu()[..., None]], -1) @ w1.double().cpu().t() + b1.double().cpu(), 0.01)
h = torch.nn.functional.leaky_relu(h @ w2.double().cpu().t() + b2.double().cpu(), 0.01)
ref = h @ w3.double().cpu().t() + b3.double().cpu()
for o, label in zip(outs, ("fp32 FMA", "MFMA hi/lo")):
    print(f"{label:12s} max |err| {float((o - ref).abs().max()):.3e}  (max |ref| {float(ref.abs().max()):.3f})")
