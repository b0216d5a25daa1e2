"""rocprofv3 counter CSV of tools/pmc_decode_sweep.py -> per-shape HBM bytes per launch.
FETCH_SIZE is reported in KB and counts HALF of the bytes of wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md,
HBM): bytes = counter * 1024 * 2.  The last 3 * len(shapes) skinny launches of the trace are the sweep, in order."""
import csv, glob, json, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
meta = json.load(open(os.path.join(os.path.dirname(root.rstrip("/")), "pmc_sweep_shapes.json")))
shapes = [tuple(s) for s in meta["shapes"]]
f = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "skinny_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-3 * len(shapes):]
agg = defaultdict(list)
for i, r in enumerate(rows):
    agg[shapes[i % len(shapes)]].append(float(r["Counter_Value"]) * 1024 * 2)
tab = {f"{n}x{k}": sum(v) / len(v) for (n, k), v in agg.items()}
algo = {f"{n}x{k}": n * k * 2 for (n, k) in agg}
res = {"source": "rocprofv3 --pmc FETCH_SIZE over tools/pmc_decode_sweep.py (x2 gfx950 wide-read correction)",
       "bytes_per_launch": tab, "algorithmic_bytes": algo, "ratio": {k: tab[k] / algo[k] for k in tab},
       "sweep_total_ratio": sum(tab[f"{n}x{k}"] for n, k in shapes) / sum(n * k * 2 for n, k in shapes),
       # the launches may stream more than the algorithm needs (folded adapter-down: [W_fc ; W_dn W_fc]); against the
       # reference model's weights once (meta["wbytes"]) the sweep's traffic is:
       "sweep_total_vs_algorithmic": sum(tab[f"{n}x{k}"] for n, k in shapes) / meta["wbytes"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["ratio"]), res["sweep_total_ratio"])
