# same-box interleaved A/B of the CU-partitioned cross-scene pipeline (bench.py --cu-split) against the sequential schedule
mkdir -p gpurun_out/cusplit
for i in 1 2; do
  for cfg in 0 64:balanced 64:low 96:balanced 32:balanced; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stages --no-parity --cu-split $cfg > gpurun_out/cusplit/bench_${cfg/:/_}_$i.json 2>> gpurun_out/cusplit/bench.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/cusplit/bench_*.json')):
    try:
        d=json.load(open(f)); rf=d['roofline']
        print(f, round(d['ms_per_step'],2), d['config'].get('cu_split'), round(rf['achieved'],1), rf['sclk_mhz_under_load'], rf['power_w'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 gpurun_out/cusplit/bench.err
