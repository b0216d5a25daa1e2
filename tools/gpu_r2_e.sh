#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sampling_gpu.py -q -m gpu -x > gpurun_out/e_sampling.log 2>&1; tail -25 gpurun_out/e_sampling.log
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_sampling_gpu.py > gpurun_out/e_pytest.log 2>&1; tail -5 gpurun_out/e_pytest.log
