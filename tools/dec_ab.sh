# (round 5, historical) same-box A/B of the LDS-staged k_decoder experiment against lib/libsemabs_hip_prev.so (profiles/r05_negative_results.md)
mkdir -p gpurun_out/r5i
python -m pytest tests/test_gpu_semabs3d.py tests/test_gpu_inference.py tests/test_gpu_scene.py -q -x -k "decoder or lattice or scene or process_batch or forward_vs_golden" > gpurun_out/r5i/tests.txt 2>&1; tail -3 gpurun_out/r5i/tests.txt
for i in 1 2; do
echo "--- new"; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2))"
echo "--- prev"; SEMABS_LIB_PATH=$PWD/semantic-abstraction_amd/lib/libsemabs_hip_prev.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],2))"
done
