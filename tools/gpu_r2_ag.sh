#!/bin/bash
# A/B of the decode attention + fc_out co-launch variants (MAGMA_DEC_AG): workgroup order and GEMV workgroup shape
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 0 1 2 3; do
  MAGMA_DEC_AG=$v timeout 300 python tools/decode_step_bench.py 2>&1 | tail -1
done > gpurun_out/dec_ag.txt
MAGMA_DEC_AG=3 timeout 300 python -m pytest tests/test_fullwidth_gpu.py -x -q -k "greedy or cached" 2>&1 | tail -3 >> gpurun_out/dec_ag.txt
cat gpurun_out/dec_ag.txt
