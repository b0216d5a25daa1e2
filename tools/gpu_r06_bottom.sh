#!/bin/bash
# round 6 (second session): the bottom block of the frozen LM forms its input gradient for the image-prefix rows only
# (MAGMA_BOTTOM_PREFIX_ONLY): parity tests, same-box A/B of the bf16 and fp8 steps
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest -x -q -m gpu tests/test_backward_kernels_gpu.py tests/test_fullwidth_train_gpu.py tests/test_train_gpu.py tests/test_variants_gpu.py tests/test_fulldepth_gpu.py tests/test_dp_engine_gpu.py 2>&1 | grep -E "passed|failed|rror|assert" | tail -6 > gpurun_out/r06h_pytest.log
cat gpurun_out/r06h_pytest.log
rm -f gpurun_out/r06h_step_ab.jsonl
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 all --no-train-truncate 2>gpurun_out/r06h.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss'], 'train_fp8_ms': t['full_S2048_fp8'].get('ms_per_step'), 'fp8_loss': t['full_S2048_fp8'].get('loss'), 'mem_GB': t.get('max_memory_allocated_GB')}
print(json.dumps(o))" >> gpurun_out/r06h_step_ab.jsonl; tail -2 gpurun_out/r06h.err; }
run MAGMA_BOTTOM_PREFIX_ONLY=0
run MAGMA_BOTTOM_PREFIX_ONLY=1
run MAGMA_BOTTOM_PREFIX_ONLY=0
run MAGMA_BOTTOM_PREFIX_ONLY=1
cat gpurun_out/r06h_step_ab.jsonl
