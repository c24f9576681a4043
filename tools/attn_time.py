"""Time the fused attention kernel at the bench shape: python tools/attn_time.py [n_tiles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
T, H, D = 197, 12, 768
qkv = (torch.randn(n * T, 3 * D, device="cuda") * 0.5).half()
out = torch.empty(n * T, D, device="cuda", dtype=torch.float16)
def run():
    _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(out), None, n, T, H, 64, 3 * D, 0, _lib.stream())
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"attention n={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
