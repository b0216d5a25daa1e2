#!/bin/bash
# round 5: per-kernel totals of two steady-state training steps (rocprofv3 kernel trace of tools/train_trace.py) under env knobs
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd)
TAG=${TAG:-r05}
rm -rf gpurun_out/train_trace
cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/train_trace -o t -- python $ROOT/tools/train_trace.py > $ROOT/gpurun_out/${TAG}_train_trace.log 2>&1; cd $ROOT
TOP=${TOP:-45} python tools/trace_summary.py gpurun_out/train_trace advance_pos > gpurun_out/${TAG}_train_step_kernel_trace_summary.txt 2>&1
TOP=70 python tools/trace_by_grid.py gpurun_out/train_trace advance_pos > gpurun_out/${TAG}_train_trace_by_grid.txt 2>&1 || true
rm -rf gpurun_out/train_trace
grep ms_per_step gpurun_out/${TAG}_train_trace.log; head -${HEAD:-30} gpurun_out/${TAG}_train_step_kernel_trace_summary.txt
