#!/bin/bash
# round 2, call B: attention forward variants (0 = 16-query waves, 1 = 32-query waves, 2 = + scheduling hints): parity + time
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 2 3; do
  MAGMA_ATTN_FWD=$v timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_properties_gpu.py tests/test_odd_shapes_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -q -m gpu -x -k "attention or flash or prefill or odd or gradients" > gpurun_out/b_pytest_v$v.log 2>&1; tail -3 gpurun_out/b_pytest_v$v.log
done
for v in 0 3 1 2; do
  for ab in 16; do
    MAGMA_ATTN_FWD=$v AB=$ab timeout 300 python tools/attn_bench.py 2>&1 | tail -1 | sed "s/^/variant $v: /" | tee -a gpurun_out/b_attn_bench.log
  done
done
