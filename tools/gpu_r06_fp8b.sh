#!/bin/bash
# round 6 (second session): fp8 training forward with the in-place rotary inside the quantising split pass: tests, bf16 + fp8 steps (same box)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest -x -q -m gpu tests/test_fp8_gpu.py tests/test_fullwidth_train_gpu.py tests/test_train_gpu.py 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06f_pytest.log
cat gpurun_out/r06f_pytest.log
rm -f gpurun_out/r06f_step.jsonl
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 all --no-train-truncate 2>gpurun_out/r06f.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'train_fp8_ms': t['full_S2048_fp8'].get('ms_per_step'), 'fp8_spread': t['full_S2048_fp8'].get('spread'), 'fp8_loss': t['full_S2048_fp8'].get('loss'), 'fwd_fp8': t.get('forward_only_fp8'), 'mem_GB': t.get('max_memory_allocated_GB')}
print(json.dumps(o))" >> gpurun_out/r06f_step.jsonl; tail -2 gpurun_out/r06f.err; }
run A=1
run A=2
cat gpurun_out/r06f_step.jsonl
