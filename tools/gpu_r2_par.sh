#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 > gpurun_out/par.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/par.txt
cat gpurun_out/par.txt
