"""Calibration (tuning aid, not the product): what the vendor library reaches on the ViT GEMM shapes, and the sustained MFMA issue rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
M = 256 * 197
for name, m, n, k in [("qkv", M, 2304, 768), ("out", M, 768, 768), ("fc", M, 3072, 768), ("proj", M, 768, 3072), ("big", 8192, 8192, 8192)]:
    A = torch.randn(m, k, device="cuda").half(); B = (torch.randn(n, k, device="cuda") * 0.05).half()
    for _ in range(3): C = A @ B.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C = A @ B.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"torch.matmul fp16 {name:5s} M={m} N={n} K={k}: {ms*1e3:8.1f} us {2*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
