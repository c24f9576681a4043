"""HBM traffic of the dominant kernel (the fp16 GEMM: k_gemm8 + the small-shape k_gemm_f16 launches) over a bench.py run, from rocprofv3 PMC counters, following
MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), both reported in KB, and on
gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B -> doubled before use.
   python tools/pmc_bench.py <outdir>      (writes <outdir>/gemm_pmc.json; copy it to profiles/)"""
import csv, glob, json, os, subprocess, sys, collections
out = os.path.abspath(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out, ctr)
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                    sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--no-stages"],
                   cwd="/tmp", env=env, capture_output=True, text=True)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        key = "k_gemm" if ("k_gemm_f16" in name or "k_gemm8" in name) else name.split("(")[0][:48]
        vals[key][ctr].append(float(row["Counter_Value"]))
    for g in glob.glob(os.path.join(d, "**", "*"), recursive=True):
        if os.path.isfile(g):
            os.remove(g)
scenes = max(1, len(vals.get("k_aggregate", {}).get("FETCH_SIZE", [])))       # one aggregation launch per scene
trunk = 0.0
for k, c in vals.items():
    if k == "k_gemm" or "k_layernorm" in k or "k_attention" in k:                # the ViT trunk: every GEMM, LayerNorm and attention launch
        trunk += sum(c["FETCH_SIZE"]) * 1024 * 2 + sum(c["WRITE_SIZE"]) * 1024
res = {}
for k, c in vals.items():
    n = len(c["FETCH_SIZE"])
    fetch = sum(c["FETCH_SIZE"]) / max(1, n) * 1024 * 2          # KB -> B, x2 gfx950 correction
    write = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])) * 1024
    res[k] = dict(dispatches=n, fetch_bytes_per_launch=fetch, write_bytes_per_launch=write, hbm_bytes_per_launch=fetch + write)
g = res.get("k_gemm", {})
json.dump(dict(round=6, commit=os.environ.get("SEMABS_COMMIT", "unknown"), scenes_in_run=scenes, trunk_counted_gb_per_scene=trunk / scenes / 1e9,
               kernel="k_gemm8 / k_gemm_f16 (all GEMM launches)", method="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-stages`; "
               "KB units; FETCH_SIZE doubled (gfx950 counts 128-B requests of wide reads at 64 B)", **g,
               per_kernel={k: v for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:14]}),
          open(os.path.join(out, "gemm_pmc.json"), "w"), indent=1)
print(json.dumps(g))
