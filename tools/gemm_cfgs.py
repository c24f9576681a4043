import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd
from semabs_amd import _lib
from semabs_amd.clip.vit import _gemm
M = 220 * 197
shapes = [("qkv", M, 2304, 768, 0), ("out", M, 768, 768, 2), ("fc", M, 3072, 768, 1), ("proj", M, 768, 3072, 2)]
for cfg in (0, 7, 4, 6, 0):
    _lib.call("semabs_gemm_set_config", cfg)
    line = []
    for name, m, n, k, epi in shapes:
        A = torch.randn(m, k, device="cuda").half(); B = (torch.randn(n, k, device="cuda") * 0.05).half()
        bias = torch.randn(n, device="cuda")
        C = torch.zeros(m, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
        for _ in range(3): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line.append(f"{name} {ms*1e3:6.1f}us {2*m*n*k/ms/1e9:5.0f}TF")
    print(f"cfg {cfg}: " + "  ".join(line), flush=True)
_lib.call("semabs_gemm_set_config", 0)
