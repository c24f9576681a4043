"""Interleaved A/B of GEMM configs on one shape: python tools/gemm_ab.py M N K epi cfgA cfgB ... (rounds=7)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd
from semabs_amd import _lib
from semabs_amd.clip.vit import _gemm
m, n, k, epi = (int(v) for v in sys.argv[1:5])
cfgs = [int(v) for v in sys.argv[5:]]
torch.manual_seed(0)
A = torch.randn(m, k, device="cuda").half(); B = (torch.randn(n, k, device="cuda") * 0.05).half(); bias = torch.randn(n, device="cuda")
C = torch.zeros(m, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
res = {c: [] for c in cfgs}
def setc(c):
    if c >= 2000:                      # 2000 + ablate bits on the persistent kernel
        _lib.call("semabs_gemm_set_config", 8); _lib.call("semabs_gemm_set_config", 1000 + (c - 2000))
    else:
        _lib.call("semabs_gemm_set_config", c); _lib.call("semabs_gemm_set_config", 1000)
for r in range(8):
    for c in cfgs:
        setc(c)
        for _ in range(2): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
        e1.record(); torch.cuda.synchronize()
        if r > 0: res[c].append(e0.elapsed_time(e1) / 10)
for c in cfgs:
    v = res[c]
    print(f"M={m} N={n} K={k} epi={epi} cfg{c}: median {statistics.median(v)*1e3:7.1f} us  min {min(v)*1e3:7.1f} us  -> {2*m*n*k/statistics.median(v)/1e9:6.1f} TF/s (median)", flush=True)
_lib.call("semabs_gemm_set_config", 0); _lib.call("semabs_gemm_set_config", 1000)
