"""Summarise a rocprofv3 *_kernel_stats.csv: python tools/prof_summary.py <csv> [scenes] [top]"""
import csv, sys
rows = list(csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#")))
scenes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {scenes:g} scenes = {tot/1e6/scenes:.2f} ms/scene")
print(f"{'kernel':66s} {'calls':>7s} {'ms/scene':>9s} {'avg us':>9s} {'%':>6s}")
for r in rows[:top]:
    print(f"{r['Name'][:66]:66s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6/scenes:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
