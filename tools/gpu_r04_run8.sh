#!/bin/bash
# round 4: adapter options, W8A16 for MAGMA_v2, erf-GELU passes
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_fp8_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -q -m gpu -k "options or w8a16 or gelu_erf or adapter or parallel or model or train" > gpurun_out/r04_pytest_options.log 2>&1; tail -12 gpurun_out/r04_pytest_options.log
