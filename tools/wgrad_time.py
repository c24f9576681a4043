"""Time semabs_wgrad_conv3 alone on one shape:  python tools/wgrad_time.py [D=128] [C=16] [B=8] [reps=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
g = torch.Generator(device="cuda").manual_seed(0)
dZ = torch.randn(B, D, D, D, C, device="cuda", generator=g) * 1e-3
X = torch.randn(B, D, D, D, C, device="cuda", generator=g)
sc = torch.rand(B, C, device="cuda", generator=g) + 0.5
sh = torch.randn(B, C, device="cuda", generator=g) * 0.1
dW = torch.zeros(C, 27, C, device="cuda")
scratch = torch.empty(16 << 20, device="cuda")
def run():
    _lib.call("semabs_wgrad_conv3", _lib.ptr(dZ), _lib.ptr(X), _lib.ptr(sc), _lib.ptr(sh), None, _lib.ptr(dW), B, D, D, D, C, C, 0,
              _lib.ptr(scratch), scratch.numel(), _lib.stream())
for _ in range(3):
    run()
smi = None
if os.environ.get("SMI"):
    import bench
    smi = bench.SmiSampler(0, period_s=0.02); smi.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
if smi:
    smi.stop(); print("smi:", smi.summary())
vox = B * D ** 3
print(f"wgrad_conv3 [{B}, {D}^3, {C} x {C}]: {us:8.1f} us   {2 * vox * C * 4 / us / 1e6:5.2f} TB/s of dZ + X   "
      f"{2.0 * vox * 27 * C * C * 3 / us / 1e6:7.1f} TFLOP/s (3 products)   |dW| {float(dW.norm()):.6e}")
