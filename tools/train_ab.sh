# (round 5, historical) same-box A/B of the GroupNorm-backward reductions fused into the level-0 data gradient (SEMABS_FUSE_CHANRED - a switch of that experiment,
# removed with it: profiles/r05_negative_results.md) - kept as the record of how the A/B was run
mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/r5g/train_tests.txt 2>&1; tail -3 gpurun_out/r5g/train_tests.txt
python -m pytest tests/test_gpu_semabs3d.py tests/test_gpu_train_dp.py -q -x > gpurun_out/r5g/other_tests.txt 2>&1; tail -3 gpurun_out/r5g/other_tests.txt
for i in 1 2; do
echo "--- fused"; SEMABS_FUSE_CHANRED=1 python tools/train_bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-120
echo "--- unfused"; SEMABS_FUSE_CHANRED=0 python tools/train_bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-120
done
python tools/unet_bench.py 16 5 2>/dev/null | grep exact | cut -c1-100
