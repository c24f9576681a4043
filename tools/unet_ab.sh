# same-box A/B of the UNet kernels: the library saved as lib/libsemabs_hip_prev.so against the current build
mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_semabs3d.py -q -x > gpurun_out/r5d/tests.txt 2>&1; tail -3 gpurun_out/r5d/tests.txt
for i in 1 2; do
echo "--- current"; python tools/unet_bench.py 16 5 2>/dev/null | grep exact
echo "--- previous"; SEMABS_LIB_PATH=$PWD/semantic-abstraction_amd/lib/libsemabs_hip_prev.so python tools/unet_bench.py 16 5 2>/dev/null | grep exact
done
echo "--- train step current"; python tools/train_bench.py --steps 3 --warmup 1 2>/dev/null | tail -1
echo "--- train step previous"; SEMABS_LIB_PATH=$PWD/semantic-abstraction_amd/lib/libsemabs_hip_prev.so python tools/train_bench.py --steps 3 --warmup 1 2>/dev/null | tail -1
