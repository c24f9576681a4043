"""One GEMM shape, repeated (PMC collection target): python tools/gemm_one.py M N K epi [reps] [cfg] [group_m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semabs_amd
from semabs_amd import _lib
from semabs_amd.clip.vit import _gemm
m, n, k, epi = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
if len(sys.argv) > 6: _lib.call("semabs_gemm_set_config", int(sys.argv[6]))
if len(sys.argv) > 7: _lib.call("semabs_gemm_set_config", 100 + int(sys.argv[7]))
torch.manual_seed(0)
A = torch.randn(m, k, device="cuda").half(); B = (torch.randn(n, k, device="cuda") * 0.05).half(); bias = torch.randn(n, device="cuda")
C = torch.zeros(m, n, device="cuda", dtype=torch.float16 if epi in (0, 1) else torch.float32)
for _ in range(reps): _gemm(A, B, C, bias, m, n, k, k, k, n, epi)
torch.cuda.synchronize()
