#!/bin/bash
# round 6, first integrated check: kernel parity of the new attention entry points, the training-parity tests that run them, then the step A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -q -x -m gpu tests/test_backward_kernels_gpu.py -k "attention or rotary" 2>&1 | tail -8
timeout 1500 python -m pytest -q -x -m gpu tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py tests/test_variants_gpu.py --durations=8 2>&1 | tail -25
rm -f gpurun_out/r06_step_ab.jsonl
bash tools/gpu_r06_step_ab.sh
