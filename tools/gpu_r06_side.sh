#!/bin/bash
# round 6 (second session): the trunk's parameter-gradient branch on a second stream (MAGMA_TRUNK_SIDE_STREAM): tests, same-box A/B; AdamW forms
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -x -q -m gpu tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py tests/test_dp_engine_gpu.py tests/test_optimizer_gpu.py 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r06e_pytest.log
cat gpurun_out/r06e_pytest.log
rm -f gpurun_out/r06e_step_ab.jsonl gpurun_out/r06e_adamw.jsonl
for v in 0 1 2; do MAGMA_ADAMW_VARIANT=$v python tools/adamw_bench.py >> gpurun_out/r06e_adamw.jsonl 2>gpurun_out/r06e.err; done
cat gpurun_out/r06e_adamw.jsonl
run() { env "$@" timeout 900 python bench.py --train-only --train-steps 4 --train-warmup 2 --no-cpu-baseline --fp8 off --no-train-truncate 2>gpurun_out/r06e.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
o = {'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss'], 'mem_GB': t.get('max_memory_allocated_GB')}
print(json.dumps(o))" >> gpurun_out/r06e_step_ab.jsonl; tail -2 gpurun_out/r06e.err; }
run MAGMA_TRUNK_SIDE_STREAM=0
run MAGMA_TRUNK_SIDE_STREAM=1
run MAGMA_TRUNK_SIDE_STREAM=0
run MAGMA_TRUNK_SIDE_STREAM=1
cat gpurun_out/r06e_step_ab.jsonl
