#!/bin/bash
# kernel trace of the steady-state training step (B = 16, S = 2048), totals per kernel and per (kernel, grid)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/tt; STEPS=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o t -- python tools/train_trace.py > gpurun_out/train_trace.log 2>&1
{ TOP=40 python tools/trace_summary.py /tmp/tt advance_pos; python tools/trace_by_grid.py /tmp/tt advance_pos; } > gpurun_out/train_trace_summary.txt 2>&1
tail -2 gpurun_out/train_trace.log; cat gpurun_out/train_trace_summary.txt
