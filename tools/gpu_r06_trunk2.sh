#!/bin/bash
# round 6 (second session): transpose + BatchNorm parameter gradients in one pass (MAGMA_BN_GRAD_FUSED), aligned 1x1 forward operands without a copy
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_backward_kernels_gpu.py tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py tests/test_fullwidth_gpu.py 2>&1 | tail -15 > gpurun_out/r06c_trunk_pytest.log
tail -3 gpurun_out/r06c_trunk_pytest.log
rm -f gpurun_out/r06c_step_ab.jsonl
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 off --no-train-truncate 2>gpurun_out/r06c_ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss'], 'mem_GB': t.get('max_memory_allocated_GB')}
print(json.dumps(o))" >> gpurun_out/r06c_step_ab.jsonl; tail -2 gpurun_out/r06c_ab.err; }
run MAGMA_BN_GRAD_FUSED=0
run MAGMA_BN_GRAD_FUSED=1
run MAGMA_BN_GRAD_FUSED=0
run MAGMA_BN_GRAD_FUSED=1
cat gpurun_out/r06c_step_ab.jsonl
TAG=r06c HEAD=30 bash tools/gpu_r05_train_trace.sh
