"""Time the fused attention kernel at the bench shape, new (DMA staging + transposing V reads) vs the round-2 kernel:
   python tools/attn_time.py [n_tiles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2448
T, H, D = 197, 12, 768
qkv = (torch.randn(n * T, 3 * D, device="cuda") * 0.5).half()
out = torch.empty(n * T, D, device="cuda", dtype=torch.float16)
out2 = torch.empty_like(out)
def run(flag, o=out):
    _lib.call("semabs_attention", _lib.ptr(qkv), _lib.ptr(o), None, n, T, H, 64, 3 * D, flag, _lib.stream())
run(0, out); run(2, out2); torch.cuda.synchronize()
print("kernels agree:", bool(torch.equal(out, out2)), float((out.float() - out2.float()).abs().max()))
run(4, out2); torch.cuda.synchronize()
print("dma variant agrees:", bool(torch.equal(out, out2)))
for rep in range(2):
    for flag, name in ((2, "k_attention2 (reg)"), (4, "k_attention2 (dma)"), (0, "k_attention")):
        for _ in range(3): run(flag)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(flag)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gb = n * T * 4 * D * 2 / 1e9
        print(f"{name:18s} n={n}: {us:.1f} us  {gb / us * 1e3:.2f} TB/s of q+k+v+out", flush=True)
