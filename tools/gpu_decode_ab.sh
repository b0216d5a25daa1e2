#!/bin/bash
# Decode A/B on one MI355X: token-step time of the full-size MAGMA_v1 graph for each knob setting given as arguments
# ("NAME=VALUE[,NAME=VALUE...]" per run; "-" = defaults), then a kernel trace of the default step summarised per (kernel, grid).
#   tools/gpu_decode_ab.sh - MAGMA_DECODE_FOLD=0
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/decode_ab.jsonl; : > $OUT
for knobs in "$@"; do
  ( [ "$knobs" != "-" ] && export $(echo $knobs | tr ',' ' '); timeout 300 python tools/decode_step_bench.py 2>/dev/null | tail -1 >> $OUT )
done
cat $OUT | cut -c1-300
cd /tmp; rm -rf $ROOT/gpurun_out/decode_trace
TRACE_MARK=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/decode_trace -o t -- python $ROOT/tools/decode_step_bench.py > $ROOT/gpurun_out/decode_trace.log 2>&1
python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/decode_trace cast_f32_bf16 > $ROOT/gpurun_out/decode_trace_by_grid.txt 2>&1
find $ROOT/gpurun_out/decode_trace -name "*.csv" -size +2M -delete
cat $ROOT/gpurun_out/decode_trace_by_grid.txt | head -16
