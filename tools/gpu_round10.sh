#!/bin/bash
# kernel trace of a short bench run (generate + 2 train steps), summarised per kernel
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py > gpurun_out/bench11.log 2>&1; tail -1 gpurun_out/bench11.log | cut -c1-1800
cd /tmp; rm -rf $ROOT/gpurun_out/tr_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/tr_bench -o t -- python $ROOT/bench.py --steps 1 --warmup 1 --train-steps 2 --no-cpu-baseline > $ROOT/gpurun_out/tr_bench.log 2>&1
python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/tr_bench > $ROOT/gpurun_out/tr_bench_summary.txt 2>&1
find $ROOT/gpurun_out/tr_bench -name "*kernel_trace.csv" -size +8M -delete
