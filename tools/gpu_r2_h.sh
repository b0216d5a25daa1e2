#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp; rm -rf $ROOT/gpurun_out/dec_trace
TRACE_MARK=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/dec_trace -o t -- python $ROOT/tools/decode_step_bench.py > $ROOT/gpurun_out/h_trace.log 2>&1
python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/dec_trace cast_f32_bf16 | tee $ROOT/gpurun_out/h_trace_by_grid.txt
find $ROOT/gpurun_out/dec_trace -name "*.csv" -size +2M -delete
