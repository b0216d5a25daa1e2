#!/bin/bash
# round 6: attention kernels without transposed images (ds_read_b64_tr_b16): parity of every MAGMA_ATTN_BWD variant, kernel timing at
# the training shape (same box, same run), optionally a kernel trace
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -q -x -m gpu tests/test_backward_kernels_gpu.py -k "attention" 2>&1 | tail -15 > gpurun_out/r06_attn_bwd_pytest.log
cat gpurun_out/r06_attn_bwd_pytest.log
for i in 1 2; do
AB=16 ABWD=${ABWD:-4,5,6} AFWD=${AFWD:-5} timeout 300 python tools/attn_bench.py >> gpurun_out/r06_attn_bench.jsonl 2> gpurun_out/r06_attn_bench.err; tail -3 gpurun_out/r06_attn_bench.err
done
cat gpurun_out/r06_attn_bench.jsonl
if [ -n "$TRACE" ]; then
  cd /tmp; export TMPDIR=/tmp
  AB=16 ABWD=${ABWD:-4,5,6} AFWD=${AFWD:-5} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_attn_trace -o t -- python $GRAFT_REPO_ROOT/tools/attn_bench.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/r06_attn_trace 2>/dev/null | head -30
fi
