# (round 5) same-box A/B of the row-linear layers on semabs_linear_rows (SEMABS_ROWS_LINEAR=0 / 1): 62.3 -> 60.0 ms
mkdir -p gpurun_out/r5j
python -m pytest tests/test_gpu_train.py tests/test_gpu_autograd_boundary.py tests/test_vool_lamb.py -q -x > gpurun_out/r5j/train_tests.txt 2>&1; tail -4 gpurun_out/r5j/train_tests.txt
for i in 1 2; do
echo "--- rows"; SEMABS_ROWS_LINEAR=1 python tools/train_bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-120
echo "--- old"; SEMABS_ROWS_LINEAR=0 python tools/train_bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-120
done
