"""Correctness + race screen + timing of the phased 256x256x64 GEMM kernel (semabs_gemm_set_config(8)) - tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semabs_amd
from semabs_amd import _lib
from semabs_amd.clip.vit import _gemm, gemm

rng = np.random.default_rng(0)
r16 = lambda *s, scale=1.0: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float16))
ok = True
for cfg in (8,):
    _lib.call("semabs_gemm_set_config", cfg)
    for (M, N, K) in [(2304, 256, 128), (4100, 768, 3072), (5000, 2304, 768), (2048, 512, 192), (9999, 1536, 768)]:
        A, B = r16(M, K), r16(N, K, scale=0.05)
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
        ref = A.double() @ B.double().T + bias.double()
        Ad, Bd, bd = A.cuda(), B.cuda(), bias.cuda()
        mx = float(ref.abs().max())
        outs = []
        for rep in range(4):
            c = torch.empty(M, N, dtype=torch.float32, device="cuda")
            gemm(Ad, Bd, c, bd, M, N, K, K, K, N, 3)
            outs.append(c.clone())
        e3 = float((outs[0].cpu().double() - ref).abs().max()) / mx
        det = all(torch.equal(outs[0], o) for o in outs[1:])
        c16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
        gemm(Ad, Bd, c16, bd, M, N, K, K, K, N, 0)
        e0 = float((c16.cpu().double() - ref).abs().max()) / mx
        gemm(Ad, Bd, c16, bd, M, N, K, K, K, N, 1)
        e1 = float((c16.cpu().double() - ref * torch.sigmoid(1.702 * ref)).abs().max()) / mx
        res = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
        c = res.cuda()
        gemm(Ad, Bd, c, bd, M, N, K, K, K, N, 2)
        e2 = float((c.cpu().double() - ref - res.double()).abs().max()) / mx
        good = e3 < 1e-5 and e0 < 2e-3 and e1 < 2e-3 and e2 < 1e-5 and det
        ok &= good
        print(f"cfg{cfg} M={M} N={N} K={K}: epi3 {e3:.1e} epi0 {e0:.1e} epi1 {e1:.1e} epi2 {e2:.1e} deterministic {det} {'OK' if good else 'FAIL'}", flush=True)
    # row-remap epilogue with M >= 2048
    n, G, T, N, K = 50, 49, 50, 768, 3072
    A, B = r16(n * G, K), r16(N, K, scale=0.02)
    pos = torch.from_numpy(rng.standard_normal((T, N)).astype(np.float32))
    out = torch.full((n * T, N), -7.0, dtype=torch.float32, device="cuda")
    gemm(A.cuda(), B.cuda(), out, None, n * G, N, K, K, K, N, 4, addend=pos.cuda(), rowmap=(G, T, 1))
    refm = (A.double() @ B.double().T).view(n, G, N) + pos[1:].double()
    got = out.cpu().view(n, T, N)
    e4 = float((got[:, 1:].double() - refm).abs().max())
    good = e4 < 1e-4 and bool((got[:, 0] == -7.0).all())
    ok &= good
    print(f"cfg{cfg} rowmap: {e4:.1e} {'OK' if good else 'FAIL'}")
print("ALL OK" if ok else "FAILURES")
M = 256 * 197
shapes = [("qkv", M, 2304, 768, 0), ("out", M, 768, 768, 2), ("fc", M, 3072, 768, 1), ("proj", M, 768, 3072, 2), ("kv11", M, 1536, 768, 3)]
for cfg in (5, 8, 5, 8):
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
        line.append(f"{name} {2*m*n*k/ms/1e9:6.0f}")
    print(f"cfg{cfg}: " + "  ".join(line) + "  TF/s", flush=True)
_lib.call("semabs_gemm_set_config", 0)
