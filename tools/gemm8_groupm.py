import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd
from semabs_amd import _lib
from semabs_amd.clip.vit import _gemm
M = 220 * 197
shapes = [("qkv", M, 2304, 768, 0), ("out", M, 768, 768, 2), ("fc", M, 3072, 768, 1), ("proj", M, 768, 3072, 2)]
bufs = []
for name, m, n, k, epi in shapes:
    A = torch.randn(m, k, device="cuda").half(); B = (torch.randn(n, k, device="cuda") * 0.05).half()
    bias = torch.randn(n, device="cuda")
    C = torch.zeros(m, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
    bufs.append((A, B, C, bias))
for rep in range(2):
    for gm in (1, 2, 4, 8, 16, 32):
        _lib.call("semabs_gemm_set_config", 100 + gm)
        line = []
        for (name, m, n, k, epi), (A, B, C, bias) in zip(shapes, bufs):
            for _ in range(2): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line.append(f"{name} {2*m*n*k/ms/1e9:5.0f}")
        print(f"group_m {gm:2d}: " + "  ".join(line), flush=True)
_lib.call("semabs_gemm_set_config", 108)
