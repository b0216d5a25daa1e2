#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 > gpurun_out/bench_quick.txt
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/bench_quick.txt
cat gpurun_out/bench_quick.txt | cut -c1-200
