#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/i_tests.log 2>&1; tail -4 gpurun_out/i_tests.log
timeout 300 python tools/decode_step_bench.py 2>&1 | tail -1 | tee -a gpurun_out/i_decode.log
timeout 600 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline --fp8 off 2>&1 | tail -1 | cut -c1-700 | tee gpurun_out/i_bench.log
