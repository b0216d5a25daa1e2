#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 0 1 3; do MAGMA_MEGA_DBG=$d timeout 200 python tools/decode_step_bench.py 2>&1 | tail -1 | tee -a gpurun_out/k_decode.log; done
