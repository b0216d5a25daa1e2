#!/bin/bash
# gemm128 with asm LDS-DMA: GEMM / conv / model tests, then phase timing of a generate call and the bench headline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_model_gpu.py tests/test_odd_shapes_gpu.py tests/test_fullwidth_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python tools/phase_timing.py 2>&1 | tail -2
timeout 900 python bench.py --no-cpu-baseline --train-steps 1 --fp8 off 2>&1 | tail -1 | cut -c1-330; } > gpurun_out/g128.txt 2>&1
cat gpurun_out/g128.txt
