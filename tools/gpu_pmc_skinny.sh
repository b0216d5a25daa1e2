#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the decode GEMV at the fc_in shape (separate PMC passes, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_sk_$c
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sk_$c -o t -- python $ROOT/tools/pmc_skinny.py > $ROOT/gpurun_out/pmc_sk_$c.log 2>&1
  f=$(find $ROOT/gpurun_out/pmc_sk_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "skinny" in r["Kernel_Name"]]
v=[float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], "launches", len(v), "mean per launch", sum(v)/max(1,len(v)))
PY
  cp "$f" $ROOT/gpurun_out/pmc_sk_$c.csv 2>/dev/null
done
