#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/kbench.jsonl
{ timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py -x -q -k "fp8 or gemm" 2>&1 | tail -3
timeout 600 python tools/kbench.py fp8tile 2>&1 | grep fp8tile; } > gpurun_out/fp8tile.txt 2>&1
cat gpurun_out/fp8tile.txt
