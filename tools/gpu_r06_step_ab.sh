#!/bin/bash
# round 6: same-box A/B of the training step: attention without transposed images (default) against the round-5 path (MAGMA_ATTN_TR=0)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 ${FP8:-off} --no-train-truncate 2>gpurun_out/r06_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss'], 'mem_GB': t.get('max_memory_allocated_GB')}
if 'full_S2048_fp8' in t: o['train_fp8_ms'] = t['full_S2048_fp8'].get('ms_per_step')
print(json.dumps(o))" >> gpurun_out/r06_step_ab.jsonl; tail -2 gpurun_out/r06_ab.err; }
run MAGMA_ATTN_TR=0
run MAGMA_ATTN_TR=1
run MAGMA_ATTN_TR=0
run MAGMA_ATTN_TR=1
cat gpurun_out/r06_step_ab.jsonl
