#!/bin/bash
# round 4: same-box A/B of this round's default changes on the training step and the eval forward (B = 16, S = 2048)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --train-steps 4 --no-cpu-baseline --fp8 off --no-variants --no-train-truncate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
print(json.dumps({'knobs': '$*', 'tokens_per_s': d['value'], 'token_step_ms': d['roofline']['token_step']['ms'], 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'train_min_ms': t['full_S2048']['spread']['min_ms']}))" >> gpurun_out/r04_defaults_ab.jsonl; }
run A=default
run MAGMA_G256_SPLITK=0
run MAGMA_PREFILL_CAT=0
run MAGMA_G256_SPLITK=0 MAGMA_PREFILL_CAT=0 MAGMA_DECODE_FOLD=0
run A=default
cat gpurun_out/r04_defaults_ab.jsonl
