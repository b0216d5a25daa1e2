#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 -k "decode or model or generate or prefill or smoke" 2>&1 | tail -30 > gpurun_out/pytest_dec.log
grep -E "passed|failed|error" gpurun_out/pytest_dec.log
timeout 600 python bench.py --steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline > gpurun_out/bench3.log 2>&1
tail -c 1500 gpurun_out/bench3.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_train -o train -- python $ROOT/bench.py --steps 1 --warmup 0 --train-steps 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_train.log 2>&1
cd $ROOT
find gpurun_out/prof_train -name "*kernel_trace.csv" -exec rm {} \;
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot/1e6)
    for r in rows[:32]:
        print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
