mkdir -p gpurun_out/r5n
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd_boundary.py tests/test_gpu_train_dp.py -q 2>&1 | grep -E "passed|failed|^E  " | head -6 | cut -c1-220
for i in 1 2; do
  echo "--- step"; python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
done
python tools/train_calls.py "semabs_chan_reduce" 8 2>/dev/null | head -12
