#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in gemm_skinny; do
  echo "=== $k" >> gpurun_out/pytest_kernels2.log
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 180 -k "$k" 2>&1 | tail -15 >> gpurun_out/pytest_kernels2.log
done
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 300 2>&1 | tail -80 > gpurun_out/pytest_model.log
grep -E "passed|failed|error" gpurun_out/pytest_kernels2.log gpurun_out/pytest_model.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench1.log 2>&1
tail -3 gpurun_out/bench1.log
