#!/bin/bash
# round 2, call A: full GPU suite (new full-width / DP / optimizer / variants tests) + short default bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/a_pytest.log 2>&1; tail -40 gpurun_out/a_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --train-steps 1 > gpurun_out/a_bench.log 2>&1; tail -1 gpurun_out/a_bench.log | cut -c1-1500
