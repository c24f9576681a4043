#!/bin/bash
# A/B of bench.py variants inside one gpurun call: tools/bench_ab.sh "<args A>" "<args B>" [repeats]
A="$1"; B="$2"; N="${3:-2}"
for r in $(seq 1 $N); do
  for v in "$A" "$B"; do
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline $v 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[$v]', round(d['value'], 3), 'scenes/s', round(d['ms_per_step'], 1), 'ms', round(d['roofline']['achieved']), 'TF', round(d['roofline']['avg_launch_us'], 1), 'us/launch')"
  done
done
