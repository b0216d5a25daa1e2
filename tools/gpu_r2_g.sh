#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -x > gpurun_out/g_tests.log 2>&1; tail -4 gpurun_out/g_tests.log
for cfg in "0 0" "16 0" "32 0" "32 12" "48 16" "64 16" "96 24"; do set -- $cfg; MAGMA_DECODE_PREFETCH_MB=$1 MAGMA_DECODE_PREFETCH2_MB=$2 timeout 300 python tools/decode_step_bench.py 2>&1 | tail -1 | tee -a gpurun_out/g_decode.log; done
