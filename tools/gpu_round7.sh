#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > gpurun_out/pytest_all_gpu.log
grep -E "passed|failed|error|Error" gpurun_out/pytest_all_gpu.log | head -20
timeout 600 python train.py --config MAGMA_v1 --synthetic_steps 3 --micro_batch 8 --grad_accum 2 > gpurun_out/train_smoke.log 2>&1
tail -8 gpurun_out/train_smoke.log | cut -c1-300
