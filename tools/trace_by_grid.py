"""Per (kernel, grid size) average durations from a rocprofv3 kernel trace, launches after the last marker kernel only."""
import csv, glob, os, sys
from collections import defaultdict
root, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    if marker:
        last = max((i for i, r in enumerate(rows) if marker in r["Kernel_Name"]), default=-1)
        rows = rows[last + 1:]
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]
        k = (n, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
    tot = sum(v[0] for v in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"== {len(rows)} launches, busy {tot/1e6:.3f} ms, span {span/1e6:.3f} ms")
    for (n, g), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", 25))]:
        print(f"{n:50s} grid={g:>9s} calls={c:5d} total_ms={t/1e6:8.3f} avg_us={t/c/1e3:8.2f}")
