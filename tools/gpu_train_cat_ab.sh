#!/bin/bash
# round 4: [W_out | W_up] as one GEMM in the TRAINING forward (train_engine._cat_out_up): parity tests, then a same-box A/B of the step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest -q -x -m gpu tests/test_backward_kernels_gpu.py tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py \
    tests/test_variants_gpu.py "tests/test_fulldepth_gpu.py::test_training_engine_loss_at_28_blocks_s2048" 2>&1 | tail -15 > gpurun_out/r04_train_cat_pytest.log
cat gpurun_out/r04_train_cat_pytest.log
run() { env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --train-steps 5 --no-cpu-baseline --fp8 off --no-variants --no-train-truncate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
print(json.dumps({'knobs': '$*', 'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'spread': t['full_S2048']['spread'], 'loss': t['full_S2048']['loss']}))" >> gpurun_out/r04_train_cat_ab.jsonl; }
rm -f gpurun_out/r04_train_cat_ab.jsonl
run A=default
run MAGMA_TRAIN_CAT=0
run A=default
run MAGMA_TRAIN_CAT=0
cat gpurun_out/r04_train_cat_ab.jsonl
