"""FETCH_SIZE / WRITE_SIZE / L2 hit per (shape, raster) of tools/gemm_raster.py: one rocprofv3 --pmc pass per counter group over
`gemm_raster.py seq`; the i-th k_gemm8 dispatch is the i-th entry of gemm_raster.sequence().   python tools/pmc_seq.py <outdir>"""
import csv, glob, json, os, subprocess, sys, collections
out = os.path.abspath(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp", SEMABS_TUNE_LIB="1")
sys.argv = [sys.argv[0]]
PASSES = [["FETCH_SIZE", "GRBM_GUI_ACTIVE"], ["WRITE_SIZE", "TCC_HIT", "TCC_MISS"]]
M = int(os.environ.get("PROBE_M", 2448 * 197))
SHAPES = {"qkv": (2304, 768, 0), "fc": (3072, 768, 1), "proj": (768, 3072, 2), "out": (768, 768, 2), "kv32": (768, 768, 3)}
import re
src = open(os.path.join(root, "tools", "gemm_raster.py")).read()
ns = {}
exec(re.search(r"CONFIGS = \{.*?\n\}\n", src, re.S).group(0), ns)
seq = [(name, gm, sc) for name in SHAPES for (gm, sc) in ns["CONFIGS"][name]]
res = collections.defaultdict(dict)
for i, p in enumerate(PASSES):
    d = os.path.join(out, f"pass{i}")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *p, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                        sys.executable, os.path.join(root, "tools", "gemm_raster.py"), "seq"], cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"pass {i}: no counter file rc={r.returncode}\n{r.stderr[-800:]}")
        continue
    rows = [row for row in csv.DictReader(open(files[0])) if "k_gemm8" in row["Kernel_Name"]]
    byd = collections.defaultdict(dict)
    for row in rows:
        byd[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(byd)
    assert len(ids) == len(seq), (len(ids), len(seq))
    for did, key in zip(ids, seq):
        res[key].update(byd[did])
    for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):
        if os.path.isfile(f):
            os.remove(f)
table = []
for (name, gm, sc), c in res.items():
    n, k, epi = SHAPES[name]
    alg = M * k * 2 + n * k * 2 + M * n * (2 if epi in (0, 1) else (8 if epi == 2 else 4))
    fetch = c.get("FETCH_SIZE", 0) * 1024 * 2
    write = c.get("WRITE_SIZE", 0) * 1024
    hit = c.get("TCC_HIT", 0) / max(1.0, c.get("TCC_HIT", 0) + c.get("TCC_MISS", 0))
    table.append(dict(shape=name, group_m=gm, sc_w=sc, fetch_GB=round(fetch / 1e9, 3), write_GB=round(write / 1e9, 3), algorithmic_GB=round(alg / 1e9, 3),
                      ratio=round((fetch + write) / alg, 3), l2_hit=round(hit, 3), gui_active=c.get("GRBM_GUI_ACTIVE")))
    print(table[-1], flush=True)
json.dump(table, open(os.path.join(out, "raster_pmc.json"), "w"), indent=1)
