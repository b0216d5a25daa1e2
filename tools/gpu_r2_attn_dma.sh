#!/bin/bash
# attention kernels after the asm LDS-DMA change (counted lgkmcnt waits): parity tests, then forward / backward times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_backward_kernels_gpu.py tests/test_kernels_gpu.py tests/test_odd_shapes_gpu.py -x -q -k "attention or attn or prefill" 2>&1 | tail -3
for i in 1 2; do AB=16 timeout 600 python tools/attn_bench.py 2>&1 | grep fwd_ms; done
rm -rf /tmp/tr1; AB=16 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python tools/attn_bench.py > /dev/null 2>&1
TOP=10 python tools/trace_summary.py /tmp/tr1
} > gpurun_out/attn_asm_dma.txt 2>&1
cat gpurun_out/attn_asm_dma.txt
