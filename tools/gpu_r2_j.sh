#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_fullwidth_gpu.py -q -m gpu -x > gpurun_out/j_fullwidth.log 2>&1; tail -6 gpurun_out/j_fullwidth.log
for m in 1 0; do MAGMA_DECODE_MEGA=$m timeout 200 python tools/decode_step_bench.py 2>&1 | tail -1 | tee -a gpurun_out/j_decode.log; done
