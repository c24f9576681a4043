"""Per-kernel utilisation table from a tools/pmc_run.py summary:  python tools/pmc_table.py <pmc_summary.json> [rows]
cycles = GRBM_GUI_ACTIVE / 8 XCDs; matrix pipe = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles; VALU = SQ_INSTS_VALU x 4 cycles / 1024 / cycles (an
estimate: one issue slot per wave instruction); LDS = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles; HBM = FETCH_SIZE + WRITE_SIZE per dispatch as the counters
report them (KiB; the guide's gfx950 corrections are applied by tools/pmc_bench.py for the figures bench.py prints, not here: read this column as relative)."""
import json, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, v in d.items():
    if not isinstance(v, dict):
        continue
    g = lambda c: v.get(c, {}).get("mean_per_dispatch", 0.0)
    n = v.get("GRBM_GUI_ACTIVE", {}).get("dispatches", 0)
    cyc = g("GRBM_GUI_ACTIVE") / 8
    if cyc <= 0:
        continue
    rows.append((cyc * n, k[:56], n, cyc, g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc, g("SQ_INSTS_VALU") * 4 / 1024 / cyc, g("SQ_LDS_IDX_ACTIVE") / 256 / cyc,
                 (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e6, g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{'kernel':56s} {'n':>4s} {'kcyc/launch':>11s} {'share':>6s} {'mfma':>5s} {'valu':>5s} {'lds':>5s} {'HBM MB':>7s} {'bank cf':>7s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r[1]:56s} {r[2]:4d} {r[3] / 1e3:11.0f} {r[0] / tot:6.3f} {r[4]:5.2f} {r[5]:5.2f} {r[6]:5.2f} {r[7]:7.0f} {r[8]:7.2f}")
