mkdir -p gpurun_out/r5n
python tools/pmc_run.py gpurun_out/r5n/pmc_train "" -- python /root/repo/tools/train_bench.py --steps 2 --warmup 1 > gpurun_out/r5n/pmc_train.txt 2>&1
tail -5 gpurun_out/r5n/pmc_train.txt
python -m pytest tests/test_gpu_train_dp.py tests/test_gpu_autograd_boundary.py tests/test_gpu_dist.py -q 2>&1 | grep "passed\|failed"
