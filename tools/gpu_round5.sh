#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 -k "skinny or decode or model or generate or prefill or smoke" 2>&1 | tail -40 > gpurun_out/pytest_dec2.log
grep -E "passed|failed|error|Error" gpurun_out/pytest_dec2.log | head
timeout 600 python bench.py --steps 3 --warmup 1 --train-steps 0 --cpu-seconds 15 > gpurun_out/bench4.log 2>&1
tail -c 2500 gpurun_out/bench4.log
