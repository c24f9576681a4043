"""GEMM micro-benchmark on the ViT shapes (tuning aid): python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import semabs_amd
from semabs_amd import _lib
from semabs_amd.clip.vit import _gemm

M = 256 * 197
shapes = [("qkv", M, 2304, 768, 0), ("out", M, 768, 768, 2), ("fc", M, 3072, 768, 1), ("proj", M, 768, 3072, 2), ("kv11", M, 1536, 768, 3),
          ("patch", 256 * 196, 768, 768, 3), ("vjp", 4096, 3072, 768, 3), ("cls", 256, 768, 768, 3)]
torch.manual_seed(0)
import itertools
shapes = shapes[:5]
for cfg, gm in [(5, 0), (5, 0)]:
    _lib.call("semabs_gemm_set_config", cfg)
    _lib.call("semabs_gemm_set_config", 1000 + gm)
    print("ablate", gm)
    for name, m, n, k, epi in shapes:
        A = (torch.randn(m, k, device="cuda") * 1.0).half(); B = (torch.randn(n, k, device="cuda") * 0.05).half()
        bias = torch.randn(n, device="cuda")
        C = torch.zeros(m, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
        for _ in range(3): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ref = (A[:64].float() @ B.float().T + bias)
        if epi == 3:
            err = (C[:64] - ref).abs().max().item() / ref.abs().max().item()
        else:
            err = float("nan")
        print(f"cfg{cfg} {name:6s} M={m:6d} N={n:5d} K={k:5d} epi={epi}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TF/s  relerr {err:.1e}", flush=True)
_lib.call("semabs_gemm_set_config", 0)
