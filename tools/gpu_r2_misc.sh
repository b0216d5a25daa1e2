#!/bin/bash
# sanity at the end of the round: odd-shape sweep, MAGMA_v2 bench line, odd-batch training
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/robustness_sweep.py 2>&1 | tail -3 > gpurun_out/misc.txt
timeout 900 python bench.py --config MAGMA_v2 --no-cpu-baseline --fp8 off 2>&1 | tail -1 > gpurun_out/bench_v2.json
cut -c1-300 gpurun_out/bench_v2.json >> gpurun_out/misc.txt
timeout 600 python tools/train_odd_batch.py 2>&1 | tail -3 >> gpurun_out/misc.txt
cat gpurun_out/misc.txt
