#!/bin/bash
# round 5: same-box A/B of the training step / forward / generate under environment knobs given as arguments ("K=V K2=V2" per run)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
OUT=gpurun_out/${AB_OUT:-r05_step_ab.jsonl}; rm -f $OUT
run() { env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --train-steps 4 --no-cpu-baseline --fp8 ${FP8ARG:-off} --no-variants --no-train-truncate 2>gpurun_out/r05_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
print(json.dumps({'knobs': '$*', 'tokens_per_s': round(d['value'], 1), 'forward_only_ms': round(t['forward_only']['ms'], 2), 'train_ms': round(t['full_S2048']['ms_per_step'], 2), 'min_ms': round(t['full_S2048']['spread']['min_ms'], 2), 'loss': t['full_S2048']['loss'], 'fwd_fp8_ms': (t.get('forward_only_fp8') or {}).get('ms'), 'fwd_fp8_loss': (t.get('forward_only_fp8') or {}).get('loss_fp8'), 'train_fp8_ms': (t.get('full_S2048_fp8') or {}).get('ms_per_step'), 'gen_fp8': (d.get('generate_fp8') or {}).get('value'), 'max_memory_allocated_GB': round(t.get('max_memory_allocated_GB') or 0, 1), 'fwd_fp8_mx_ms': (t.get('forward_only_fp8_mx') or {}).get('ms'), 'fwd_fp8_mx_loss': (t.get('forward_only_fp8_mx') or {}).get('loss_fp8')}))" >> $OUT; }
for cfg in "$@"; do run $cfg; done
cat $OUT; tail -2 gpurun_out/r05_ab.err
