#!/bin/bash
# power / clock of the GPU while the production QKV GEMM runs back to back on random vs zero operands (rocm-smi sampled next to a python loop)
for kind in random zeros; do
  python - "$kind" > /dev/null 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, semabs_amd
from semabs_amd.clip.vit import gemm
M, N, K = 2448 * 197, 2304, 768
A = torch.randn(M, K, device="cuda").half(); B = (torch.randn(N, K, device="cuda") * 0.05).half(); bias = torch.randn(N, device="cuda")
if sys.argv[1] == "zeros": A.zero_(); B.zero_(); bias.zero_()
C = torch.empty(M, N, device="cuda", dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(50): gemm(A, B, C, bias, M, N, K, K, K, N, 0)
    torch.cuda.synchronize()
PY
  P=$!
  sleep 6
  for i in 1 2 3 4 5; do
    echo -n "QKV GEMM, $kind operands: "; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/GPU\[0\]//' | tr '\n' ';'; echo
    sleep 1
  done
  wait $P
done
