# same-box A/B: GroupNorm-backward sums from the weight-gradient pass (SEMABS_WGRAD_GN=1) vs semabs_chan_reduce (0)
mkdir -p gpurun_out/r5n
python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -4 | cut -c1-220
for i in 1 2; do for v in 1 0; do
  echo "--- step WGRAD_GN=$v"; SEMABS_WGRAD_GN=$v python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
done; done
SEMABS_WGRAD_GN=1 python tools/train_calls.py "" 16 2>/dev/null | head -1
