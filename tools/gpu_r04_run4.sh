#!/bin/bash
# round 4, fourth GPU pass: training step per (kernel, grid) -- where the CLIP trunk's small GEMMs go
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
export TOP=70; cd /tmp; rm -rf $ROOT/gpurun_out/train_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/train_trace -o t -- python $ROOT/tools/train_trace.py > /dev/null 2>&1
python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/train_trace advance_pos > $ROOT/gpurun_out/r04_train_trace_by_grid.txt 2>&1
TOP=45 python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/train_trace advance_pos > $ROOT/gpurun_out/r04_train_step_kernel_trace_summary_mid.txt 2>&1
find $ROOT/gpurun_out/train_trace -name "*.csv" -size +2M -delete
head -60 $ROOT/gpurun_out/r04_train_trace_by_grid.txt
cd $ROOT; timeout 300 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "multi_token" 2>&1 | tail -3
