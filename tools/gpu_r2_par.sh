#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | grep -v Warning | tail -30 > gpurun_out/par.txt
cat gpurun_out/par.txt
