#!/bin/bash
# round 6 (second session): vectorised AdamW / sum-of-squares, statistics pass of the attention backward with 16-byte loads:
# tests, bf16 + fp8 step in one process (same box), kernel traces of both steps
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_optimizer_gpu.py tests/test_backward_kernels_gpu.py tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py tests/test_dp_nccl_gpu.py 2>&1 | tail -6 > gpurun_out/r06d_pytest.log
tail -3 gpurun_out/r06d_pytest.log
rm -f gpurun_out/r06d_step.jsonl
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 all --no-train-truncate 2>gpurun_out/r06d.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'train_fp8_ms': t['full_S2048_fp8'].get('ms_per_step'), 'fp8_spread': t['full_S2048_fp8'].get('spread'), 'fwd_fp8': t.get('forward_only_fp8'), 'mem_GB': t.get('max_memory_allocated_GB')}
print(json.dumps(o))" >> gpurun_out/r06d_step.jsonl; tail -2 gpurun_out/r06d.err; }
run A=1
run A=2
cat gpurun_out/r06d_step.jsonl
TAG=r06d HEAD=24 bash tools/gpu_r05_train_trace.sh
MAGMA_TRAIN_FP8=1 TAG=r06d_fp8 HEAD=32 bash tools/gpu_r05_train_trace.sh
