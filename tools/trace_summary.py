"""Per-kernel totals from a rocprofv3 *_kernel_trace.csv, skipping the first SKIP fraction (setup)."""
import csv, sys, glob, os
from collections import defaultdict
root = sys.argv[1]
skip_until = sys.argv[2] if len(sys.argv) > 2 else None
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if skip_until:      # keep only what was launched after the LAST kernel whose name contains the marker
        last = max((i for i, r in enumerate(rows) if skip_until in r["Kernel_Name"]), default=-1)
        rows = rows[last + 1:]
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        n = n[:70]
        agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[n][1] += 1
    tot = sum(v[0] for v in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"== {f}: {len(rows)} launches, busy {tot/1e6:.2f} ms, span {span/1e6:.2f} ms")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get('TOP', 45))]:
        print(f"{n:72s} calls={c:6d} total_ms={t/1e6:9.3f} avg_us={t/c/1e3:8.2f}")
