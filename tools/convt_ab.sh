# same-box A/B of k_convT_brick's brick depth (new: 2 in exact mode, two workgroups per CU; prev = lib/libsemabs_hip_prev.so: 4, one per CU)
L=$PWD/semantic-abstraction_amd/lib
python -m pytest tests/test_gpu_semabs3d.py -q -k "convtranspose or convT or unet or forward" 2>&1 | grep -E "passed|failed" 
for i in 1 2; do for v in "" _prev; do
  echo "--- lib$v"; SEMABS_LIB_PATH=$L/libsemabs_hip$v.so python tools/unet_bench.py 16 5 2>/dev/null | grep -i exact | cut -c1-120
  SEMABS_LIB_PATH=$L/libsemabs_hip$v.so python tools/train_bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-100
  SEMABS_LIB_PATH=$L/libsemabs_hip$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stages --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('scene ms', round(d['ms_per_step'],2))"
done; done
