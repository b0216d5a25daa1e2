#!/bin/bash
# round 6: fp8 adapter chain (config[4] "+ adapter GEMMs"): parity, then same-box A/B of the fp8 training step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest -q -x -m gpu ${TESTS:-tests/test_fullwidth_train_gpu.py tests/test_fp8_gpu.py tests/test_train_gpu.py tests/test_nfresnet_gpu.py} --durations=12 2>&1 | tail -22 | cut -c1-200
echo "wall seconds: $(( $(date +%s) - T0 ))"
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 all --no-train-truncate 2>gpurun_out/r06_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'train_ms': t['full_S2048']['ms_per_step'], 'train_fp8_ms': t['full_S2048_fp8'].get('ms_per_step'), 'fp8_spread': t['full_S2048_fp8'].get('spread'), 'fp8_loss': t['full_S2048_fp8'].get('loss'), 'fwd_fp8': t.get('forward_only_fp8'), 'err': t['full_S2048_fp8'].get('error')}
print(json.dumps(o))" >> gpurun_out/r06_fp8_step_ab.jsonl; tail -2 gpurun_out/r06_ab.err; }
rm -f gpurun_out/r06_fp8_step_ab.jsonl
run MAGMA_FP8_ADAPTERS=0
run MAGMA_FP8_ADAPTERS=1
run MAGMA_FP8_ADAPTERS=0
run MAGMA_FP8_ADAPTERS=1
cat gpurun_out/r06_fp8_step_ab.jsonl | cut -c1-600
