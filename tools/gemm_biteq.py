"""Bit-equality of the GEMM schedules on the trunk's fp16-output shapes (every schedule must reproduce round 3's PF schedule bit for bit: same fragments, same MFMA order, same fp32
epilogue order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd.clip.vit import gemm
for epi in (0, 1, 2):        # (epilogue 2 has no persistent variant in the product: the third column then equals the second)
    for (M, N, K) in [(2381, 2304, 768), (50432, 3072, 768), (2048, 768, 128), (50432, 768, 3072)]:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        A = torch.randn(M, K, device="cuda", generator=g).half()
        B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(N, device="cuda", generator=g)
        out = []
        for kern in (2 | 2048, 2 | 4096, 2):
            C = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda") if epi < 2 else torch.sin(torch.arange(M * N, device="cuda", dtype=torch.float32)).view(M, N).contiguous()
            gemm(A, B, C, bias, M, N, K, K, K, N, epi, kernel=kern)
            torch.cuda.synchronize()
            out.append(C)
        print(f"epi {epi} {M}x{N}x{K}: PF == deep {torch.equal(out[0], out[1])}  PF == persistent {torch.equal(out[0], out[2])}", flush=True)
