# (round 5, historical) same-box A/B of the two-workgroup k_wgrad3_tr experiment against lib/libsemabs_hip_prev.so (profiles/r05_negative_results.md)
mkdir -p gpurun_out/r5h
python -m pytest tests/test_gpu_train.py -q -x -k "strict or wgrad or golden_64" > gpurun_out/r5h/train_tests.txt 2>&1; tail -3 gpurun_out/r5h/train_tests.txt
for i in 1 2; do
echo "--- new"; python tools/train_bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-120
echo "--- prev"; SEMABS_LIB_PATH=$PWD/semantic-abstraction_amd/lib/libsemabs_hip_prev.so python tools/train_bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-120
done
