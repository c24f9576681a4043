# same-box A/Bs of the round-5 training-step fusions (environment switches of semabs_amd/train.py), interleaved:
#   SEMABS_WGRAD_GN=0        semabs_wgrad_conv3 + semabs_chan_reduce instead of semabs_wgrad_conv3_gn (also turns the next one off: it needs the sums first)
#   SEMABS_FUSE_GN_APPLY=0   semabs_conv3d + semabs_gn_bwd_apply instead of semabs_conv3d_gnbwd
mkdir -p gpurun_out/r5n
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd_boundary.py tests/test_gpu_train_dp.py -q 2>&1 | grep -E "passed|failed|^E  " | head -6 | cut -c1-220
for i in 1 2; do
  echo "--- default";              python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
  echo "--- SEMABS_FUSE_GN_APPLY=0"; SEMABS_FUSE_GN_APPLY=0 python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
  echo "--- SEMABS_WGRAD_GN=0";    SEMABS_WGRAD_GN=0 python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-120
done
