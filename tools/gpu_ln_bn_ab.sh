#!/bin/bash
# round 4: LayerNorm with four rows per workgroup + BatchNorm parameter gradients with four rows in flight: tests, then a same-box A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest -q -x -m gpu tests/test_kernels_gpu.py -k "layernorm" tests/test_backward_kernels_gpu.py tests/test_train_gpu.py 2>&1 | tail -5 > gpurun_out/r04_ln_bn_pytest.log
cat gpurun_out/r04_ln_bn_pytest.log
run() { env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --train-steps 5 --no-cpu-baseline --fp8 off --no-variants --no-train-truncate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
print(json.dumps({'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss']}))" >> gpurun_out/r04_ln_rows_ab.jsonl; }
rm -f gpurun_out/r04_ln_rows_ab.jsonl
run A=default
run MAGMA_LN_ROWS=1
run A=default
run MAGMA_LN_ROWS=1
cat gpurun_out/r04_ln_rows_ab.jsonl
