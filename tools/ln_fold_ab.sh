mkdir -p gpurun_out/r5c
python -m pytest tests/test_gpu_gemm.py -q -x -k "layernorm or trunk_with" -rP > gpurun_out/r5c/ln_tests.txt 2>&1; tail -4 gpurun_out/r5c/ln_tests.txt
python -m pytest tests/test_gpu_headline.py tests/test_gpu_relevancy.py -q -rP > gpurun_out/r5c/rel_tests.txt 2>&1; tail -4 gpurun_out/r5c/rel_tests.txt
for i in 1 2; do
SEMABS_LN_FOLD=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity > gpurun_out/r5c/bench_fold_$i.json 2>> gpurun_out/r5c/bench.err
SEMABS_LN_FOLD=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity > gpurun_out/r5c/bench_nofold_$i.json 2>> gpurun_out/r5c/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/bench_*.json')):
    try:
        d=json.load(open(f)); rf=d['roofline']
        print(f, round(d['ms_per_step'],2), round(rf['achieved'],1), rf['sclk_mhz_under_load'], rf['power_w'])
        for k,v in rf['per_shape'].items():
            if v['launches']>=40 and v['avg_us']>300: print('    ',k, v['avg_us'], v['tflops'])
    except Exception as e: print(f, 'ERR', e)
PY
