#!/bin/bash
# round 2, call D: kernel trace of steady-state training steps (per-kernel totals per step)
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp; rm -rf $ROOT/gpurun_out/train_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/train_trace -o t -- python $ROOT/tools/train_trace.py > $ROOT/gpurun_out/train_trace.log 2>&1
tail -2 $ROOT/gpurun_out/train_trace.log
TOP=60 python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/train_trace advance_pos > $ROOT/gpurun_out/train_trace_summary.txt 2>&1
find $ROOT/gpurun_out/train_trace -name "*.csv" -size +2M -delete
cat $ROOT/gpurun_out/train_trace_summary.txt
