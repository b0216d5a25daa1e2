#!/bin/bash
# first GPU contact: kernel parity tests (one process per group so a faulting
# kernel cannot take the rest down), then micro-benchmarks.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt
for k in gemm_dense transpose_detecting gemm_epilogue conv3x3 gemm_skinny tile_roundtrip layernorm embedding rotary_split online_softmax decode_attention argmax avgpool build_labels cross_entropy errors_are_loud; do
  echo "=== $k" >> gpurun_out/pytest_kernels.log
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 180 -k "$k" 2>&1 | tail -40 >> gpurun_out/pytest_kernels.log
done
grep -E "^===|passed|failed|error" gpurun_out/pytest_kernels.log
timeout 900 python tools/kbench.py all > gpurun_out/kbench.log 2>&1
tail -5 gpurun_out/kbench.log
