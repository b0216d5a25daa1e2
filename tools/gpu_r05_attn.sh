#!/bin/bash
# round 5: the 32-row-wave attention backward (attention_bwd32.hip): parity of every MAGMA_ATTN_BWD variant, kernel timing at the
# training shape, then a same-box A/B of the training step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -q -x -m gpu tests/test_backward_kernels_gpu.py -k "attention" 2>&1 | tail -15 > gpurun_out/r05_attn_bwd_pytest.log
cat gpurun_out/r05_attn_bwd_pytest.log
AB=16 timeout 300 python tools/attn_bench.py > gpurun_out/r05_attn_bench.jsonl 2> gpurun_out/r05_attn_bench.err; tail -3 gpurun_out/r05_attn_bench.err
AB=16 timeout 300 python tools/attn_bench.py >> gpurun_out/r05_attn_bench.jsonl 2>> gpurun_out/r05_attn_bench.err
cat gpurun_out/r05_attn_bench.jsonl
if grep -q failed gpurun_out/r05_attn_bwd_pytest.log; then echo "parity failed: no step A/B"; exit 1; fi
run() { env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --train-steps 4 --no-cpu-baseline --fp8 off --no-variants --no-train-truncate 2>gpurun_out/r05_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
print(json.dumps({'knobs': '$*', 'tokens_per_s': d['value'], 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss']}))" >> gpurun_out/r05_attn_bwd_step_ab.jsonl; }
rm -f gpurun_out/r05_attn_bwd_step_ab.jsonl
run MAGMA_ATTN_BWD=0
run MAGMA_ATTN_BWD=2
run MAGMA_ATTN_BWD=4
cat gpurun_out/r05_attn_bwd_step_ab.jsonl; tail -3 gpurun_out/r05_ab.err
