"""Condense rocprofv3 output (kernel stats CSV, kernel trace CSV, PMC counter CSVs)
into a small text summary that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n if len(n) < 90 else n[:87] + "..."


for f in glob.glob(os.path.join(root, "prof_trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(f"{short(r.get('Name','')):90s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
              f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for tag in ("prof_fetch", "prof_write"):
    for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        print("== counters:", f)
        agg = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (short(r.get("Kernel_Name", "")), r.get("Counter_Name", ""))
            agg[k][0] += float(r.get("Counter_Value", 0) or 0)
            agg[k][1] += 1
        for (kn, cn), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:20]:
            print(f"{kn:90s} {cn} sum={v:.4g} launches={n} per_launch={v/max(n,1):.4g}")
