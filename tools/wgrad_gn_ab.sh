mkdir -p gpurun_out/r5n
python -m pytest tests/test_gpu_semabs3d.py -q -x -k "conv_brick_kernel or precision or conv3d" 2>&1 | tail -3 | cut -c1-220
python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -2 | cut -c1-220
for i in 1 2; do
  echo "--- step"; python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
done
python tools/train_calls.py "semabs_conv3d" 6 2>/dev/null | head -5
