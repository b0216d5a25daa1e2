#!/bin/bash
# round 6: the CLIP RN50x16 inference trunk at B = 8, 224^2 (2.6 ms of the 88-ms generate call): per-launch sequence and per-(kernel, grid) totals
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/enc_trace; cd /tmp
PHASE=encoder ITERS=6 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/enc_trace -o t -- python $ROOT/tools/prefill_prof.py > /dev/null 2>&1
cd $ROOT
python tools/trace_by_grid.py gpurun_out/enc_trace cast_f32_bf16 > gpurun_out/r06_encoder_trace_by_grid.txt 2>&1
python tools/trace_sequence.py gpurun_out/enc_trace cast_f32_bf16 0 400 > gpurun_out/r06_encoder_layer_sequence.txt 2>&1
find gpurun_out/enc_trace -name "*.csv" -size +2M -delete
head -40 gpurun_out/r06_encoder_trace_by_grid.txt
