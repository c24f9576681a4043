# same-box A/B of k_wgrad3_tr: new (loader / multiplier waves) vs prev (lib/libsemabs_hip_prev.so = HEAD before the change)
mkdir -p gpurun_out/r5n
L=$PWD/semantic-abstraction_amd/lib
for i in 1 2 3; do python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -1; done
python -m pytest tests/test_gpu_train_dp.py tests/test_gpu_autograd_boundary.py tests/test_gpu_dist.py -q 2>&1 | tail -2
for v in "" _prev; do
  echo "--- lib$v"; SEMABS_LIB_PATH=$L/libsemabs_hip$v.so python tools/train_calls.py semabs_wgrad_conv3 24 2>/dev/null | tail -25 | cut -c1-110
done
for i in 1 2; do for v in "" _prev; do
  echo "--- step lib$v"; SEMABS_LIB_PATH=$L/libsemabs_hip$v.so python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
done; done
