#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu -x > gpurun_out/f_train.log 2>&1; tail -15 gpurun_out/f_train.log
for kc in 4 8 16; do MAGMA_SKINNY2_KC=$kc timeout 300 python tools/decode_step_bench.py 2>&1 | tail -1 | tee -a gpurun_out/f_decode.log; done
