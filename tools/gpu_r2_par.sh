#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{ timeout 600 python tools/kbench.py shortk 2>&1 | grep shortk
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_train_gpu.py tests/test_backward_kernels_gpu.py tests/test_model_gpu.py tests/test_odd_shapes_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 python bench.py --no-cpu-baseline --steps 2 2>&1 | tail -1; } > gpurun_out/par.txt 2>&1
cut -c1-200 gpurun_out/par.txt
