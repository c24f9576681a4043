mkdir -p gpurun_out/r5m
echo "--- new"; python tools/attn_time.py 2>/dev/null | grep "agree\|k_attention "
echo "--- prev"; SEMABS_LIB_PATH=$PWD/semantic-abstraction_amd/lib/libsemabs_hip_prev.so python tools/attn_time.py 2>/dev/null | grep "agree\|k_attention "
python -m pytest tests/test_gpu_relevancy.py tests/test_vit_l14.py tests/test_gpu_headline.py tests/test_gpu_gemm.py -q -k "not gemm_kernels_all" 2>&1 | grep "passed\|failed"
for i in 1 2; do
echo new; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2))"
echo prev; SEMABS_LIB_PATH=$PWD/semantic-abstraction_amd/lib/libsemabs_hip_prev.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2))"
done
