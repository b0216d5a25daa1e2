#!/bin/bash
# round 4, third GPU pass: timing ablations of the pipelined attention forward (ablation library), decode block variants, changed tests
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
for a in 0 1 2 3 4 5 0; do MAGMA_HIP_LIB=$ROOT/magma_amd/libmagma_hip_abl.so MAGMA_ATTN_ABL=$a AB=16 timeout 120 python tools/attn_bench.py 2>/dev/null | sed "s/^/{\"abl\": $a} /" >> gpurun_out/r04_attn_fwd_ablations.txt; done
for v in 4 5 6 3 5 4; do MAGMA_ATTN_FWD=$v AB=16 timeout 120 python tools/attn_bench.py 2>/dev/null | sed "s/^/{\"fwd_variant\": $v} /" >> gpurun_out/r04_attn_fwd_ab.txt; done
cat gpurun_out/r04_attn_fwd_ablations.txt; tail -6 gpurun_out/r04_attn_fwd_ab.txt
MAGMA_ATTN_FWD=5 timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_odd_shapes_gpu.py -q -m gpu -x -k "attn or attention or prefill or odd" > gpurun_out/r04_pytest_attn_v5.log 2>&1; tail -3 gpurun_out/r04_pytest_attn_v5.log
for f in 0 2 0 2; do MAGMA_DECODE_FOLD=$f timeout 300 python tools/decode_step_bench.py 2>/dev/null | head -1 >> gpurun_out/r04_decode_fold_ab.jsonl; done
tail -4 gpurun_out/r04_decode_fold_ab.jsonl
timeout 900 python -m pytest tests/test_fullwidth_gpu.py tests/test_nfresnet_gpu.py tests/test_variants_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > gpurun_out/r04_pytest_changed.log 2>&1; tail -6 gpurun_out/r04_pytest_changed.log
