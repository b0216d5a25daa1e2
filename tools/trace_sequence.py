"""The launches of ONE layer in order (start offset, duration, gap to the previous end) from a rocprofv3 kernel trace: launches after the last
marker kernel, then launches [SKIP, SKIP + COUNT).  python tools/trace_sequence.py <dir> <marker> <skip> <count>"""
import csv, glob, os, sys
root, marker, skip, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    last = max((i for i, r in enumerate(rows) if marker in r["Kernel_Name"]), default=-1)
    rows = rows[last + 1:][skip: skip + count]
    t0, prev = int(rows[0]["Start_Timestamp"]), None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]
        print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {0 if prev is None else (s - prev) / 1e3:6.2f}  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>4s}  {n}")
        prev = e
