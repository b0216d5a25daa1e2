#!/bin/bash
# round 4, second GPU pass: attention forward A/B (chain per tile vs software-pipelined), attention tests, decode trace with the folded block
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 3 4 3 4; do MAGMA_ATTN_FWD=$v AB=16 timeout 120 python tools/attn_bench.py 2>/dev/null | sed "s/^/{\"fwd_variant\": $v} /" >> gpurun_out/r04_attn_fwd_ab.txt; done
cat gpurun_out/r04_attn_fwd_ab.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_odd_shapes_gpu.py tests/test_fullsize_properties_gpu.py tests/test_fullwidth_train_gpu.py tests/test_train_gpu.py -q -m gpu -x -k "attn or attention or flash or prefill or odd or properties or gradients or fp8_training or train" > gpurun_out/r04_pytest_attn.log 2>&1; tail -6 gpurun_out/r04_pytest_attn.log
cd /tmp; rm -rf $ROOT/gpurun_out/decode_trace_fold
MAGMA_DECODE_FOLD=1 TRACE_MARK=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/decode_trace_fold -o t -- python $ROOT/tools/decode_step_bench.py > /dev/null 2>&1
python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/decode_trace_fold cast_f32_bf16 > $ROOT/gpurun_out/r04_decode_fold_trace_by_grid.txt 2>&1
find $ROOT/gpurun_out/decode_trace_fold -name "*.csv" -size +2M -delete
head -12 $ROOT/gpurun_out/r04_decode_fold_trace_by_grid.txt
