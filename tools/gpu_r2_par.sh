#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_train_gpu.py tests/test_dp_engine_gpu.py tests/test_optimizer_gpu.py -x -q -s 2>&1 | grep -v Warning | tail -30 > gpurun_out/par.txt
cat gpurun_out/par.txt
