#!/bin/bash
# power / clock of the GPU while the trunk GEMMs run back to back (rocm-smi sampled next to tools/gemm_probe.py)
SEMABS_TUNE_LIB=0 python tools/gemm_probe.py deep deep deep > /dev/null 2>&1 &
P=$!
for i in $(seq 1 40); do
  if ! kill -0 $P 2>/dev/null; then break; fi
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/GPU\[0\]//' | tr '\n' ';'; echo
  sleep 1
done
wait $P
