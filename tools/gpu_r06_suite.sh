#!/bin/bash
# round 6: the whole GPU suite with its wall time and the slowest tests, then smoke()
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s); timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/${TAG:-r06}_pytest_gpu.log 2>&1
echo "suite wall seconds: $(( $(date +%s) - T0 ))" | tee -a gpurun_out/${TAG:-r06}_pytest_gpu.log; tail -30 gpurun_out/${TAG:-r06}_pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
