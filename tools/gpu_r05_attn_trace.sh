#!/bin/bash
# round 5: per-kernel times of the attention backward variants at the training shape (rocprofv3 kernel trace of tools/attn_bench.py)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd)
rm -rf gpurun_out/r05_attn_trace
cd /tmp
AB=16 ABWD=${ABWD:-0,4} timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r05_attn_trace -o t -- python $ROOT/tools/attn_bench.py > $ROOT/gpurun_out/r05_attn_trace.log 2>&1
cd $ROOT
TOP=14 python tools/trace_summary.py gpurun_out/r05_attn_trace > gpurun_out/r05_attn_trace_summary.txt 2>&1
find gpurun_out/r05_attn_trace -name "*.db" -delete 2>/dev/null; rm -rf gpurun_out/r05_attn_trace
cat gpurun_out/r05_attn_trace_summary.txt
