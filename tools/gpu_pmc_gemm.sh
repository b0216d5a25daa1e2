#!/bin/bash
# HBM traffic of the 256x256 GEMM kernels (bf16 and fp8) at the training shapes: FETCH_SIZE / WRITE_SIZE, one pass each
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_gemm_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_gemm_$c -o t -- python $ROOT/tools/kbench.py fp8tile > $ROOT/gpurun_out/pmc_gemm_$c.log 2>&1
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
root=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
out=open(root+"/pmc_gemm_summary.txt","w")
print("kbench fp8tile under rocprofv3 --pmc (M = 32768; shapes qkv N=12288 K=4096, out_proj 4096x4096, fc_in 16384x4096, fc_out 4096x16384)", file=out)
print("per (kernel, grid): mean counter value per launch; FETCH_SIZE / WRITE_SIZE are in KB, FETCH_SIZE x2 on gfx950 for the corrected figure", file=out)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(root+f"/pmc_gemm_{c}/**/*counter_collection.csv", recursive=True):
        agg=defaultdict(lambda:[0.0,0])
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]: continue
            k=(r["Kernel_Name"].replace("(anonymous namespace)::","")[:44], r.get("Grid_Size", r.get("Grid_Size_X","?")))
            agg[k][0]+=float(r["Counter_Value"]); agg[k][1]+=1
        for (kn,g),(v,n) in sorted(agg.items()):
            print(f"{c:11s} {kn:46s} grid={g:>9s} per_launch_KB={v/n:12.0f} launches={n}", file=out)
out.close(); print(open(root+"/pmc_gemm_summary.txt").read())
PY
find $ROOT/gpurun_out/pmc_gemm_* -name "*.csv" -size +4M -delete
