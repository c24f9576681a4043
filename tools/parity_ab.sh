mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_gemm.py -q -x -k "low_halves or attention_split" -rP > gpurun_out/r5f/unit.txt 2>&1; tail -12 gpurun_out/r5f/unit.txt
python -m pytest tests/test_gpu_headline.py -q -rP > gpurun_out/r5f/headline.txt 2>&1; grep "headline\|passed\|failed" gpurun_out/r5f/headline.txt
SEMABS_QK_SPLIT=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity > gpurun_out/r5f/bench_default.json 2>> gpurun_out/r5f/bench.err
SEMABS_QK_SPLIT=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stages --no-parity > gpurun_out/r5f/bench_parity.json 2>> gpurun_out/r5f/bench.err
python - <<'PY'
import json
for n in ("default","parity"):
    d=json.load(open(f"gpurun_out/r5f/bench_{n}.json")); print(n, round(d["ms_per_step"],2), round(d["roofline"]["achieved"],1))
PY
