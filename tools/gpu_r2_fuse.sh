#!/bin/bash
# training-step fusions: merged attention backward output, q^T/k^T from rotary_split, dO^T + D in one pass
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_backward_kernels_gpu.py tests/test_train_gpu.py tests/test_dp_engine_gpu.py tests/test_optimizer_gpu.py -x -q 2>&1 | tail -5
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
} > gpurun_out/fuse.txt 2>&1
cat gpurun_out/fuse.txt
