"""Collect rocprofv3 PMC counters in separate passes for one command and print per-kernel means.
   python tools/pmc_run.py <outdir> <kernel-substring> -- <command...>
Each pass is its own rocprofv3 run with --kernel-trace only (never combined with sys/hip traces)."""
import csv, glob, os, subprocess, sys, collections, json
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"],
    ["FETCH_SIZE", "GRBM_GUI_ACTIVE"],
    ["WRITE_SIZE", "TCC_HIT", "TCC_MISS"],
    ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INST_LEVEL_VMEM", "SQ_ACTIVE_INST_LDS"],
    ["TCP_TCC_READ_REQ", "TCP_TOTAL_CACHE_ACCESSES", "TCC_EA0_RDREQ", "TCC_REQ"],
]
out, pat = os.path.abspath(sys.argv[1]), sys.argv[2]
cmd = sys.argv[sys.argv.index("--") + 1:]
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
res = collections.defaultdict(lambda: collections.defaultdict(list))
for i, p in enumerate(PASSES):
    d = os.path.join(out, f"pass{i}")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *p, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + cmd,
                       cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"pass {i} produced no counter file; rc={r.returncode}\n{r.stderr[-600:]}")
        continue
    for row in csv.DictReader(open(files[0])):
        if pat in row["Kernel_Name"]:
            res[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):
        if os.path.isfile(f) and not f.endswith("counter_collection.csv"):
            os.remove(f)
summary = {}
for k, cs in res.items():
    print(k)
    summary[k] = {}
    for c, v in cs.items():
        m = sum(v) / len(v)
        summary[k][c] = {"mean_per_dispatch": m, "dispatches": len(v)}
        print(f"   {c:28s} mean/dispatch {m:16.1f}   (n={len(v)})")
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
