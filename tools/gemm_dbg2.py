"""Debug aid: one epilogue, two schedules, element-wise comparison and the pattern of differing elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd  # noqa
from semabs_amd.clip.vit import gemm
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (M, N, K) in [(2048, 768, 768), (2048, 768, 128), (2381, 2304, 768)]:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).half()
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g)
    outs = {}
    for kern in (2 | 2048, 2, 2):
        C = torch.full((M, N), float("nan"), dtype=torch.float16 if epi in (0, 1) else torch.float32, device="cuda")
        if epi == 2: C.zero_()
        gemm(A, B, C, bias, M, N, K, K, K, N, epi, kernel=kern)
        torch.cuda.synchronize()
        outs.setdefault(kern, []).append(C)
    a, b, b2 = outs[2 | 2048][0], outs[2][0], outs[2][1]
    d = (a.float() - b.float()).abs(); d[torch.isnan(d)] = 1e9
    bad = d > 1e-2
    r, c = torch.nonzero(bad, as_tuple=True)
    print(f"epi {epi} M{M} N{N} K{K}: deep vs PF differing {int(bad.sum())}; deep run1 vs run2 differing {int(((b.float() - b2.float()).abs() > 0).sum())} nan in deep {int(torch.isnan(b.float()).sum())}")
    if len(r):
        print("   rows%256:", sorted(set((r % 256).tolist()))[:40])
        print("   cols%256:", sorted(set((c % 256).tolist()))[:40])
        print("   tiles:", sorted(set(((r // 256) * 100 + c // 256).tolist()))[:20])
