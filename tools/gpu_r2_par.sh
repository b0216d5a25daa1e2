#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{ timeout 600 python tools/kbench.py shortk 2>&1 | grep shortk | grep '"tile": 256'
timeout 600 python tools/kbench.py fp8tile 2>&1 | grep fp8tile | cut -c1-330; } > gpurun_out/par.txt 2>&1
cat gpurun_out/par.txt
