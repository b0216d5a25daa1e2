#!/bin/bash
# round 4, fifth GPU pass: K-concatenated [W_out | W_up] in the prefill / forward blocks -- parity + generate timing; tr-read probe
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 tools/probes/tr_probe > gpurun_out/r04_tr_probe.txt 2>&1; head -70 gpurun_out/r04_tr_probe.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py tests/test_config1_gpu.py tests/test_fulldepth_gpu.py -q -m gpu -x -k "not training_engine_loss and not gradients" > gpurun_out/r04_pytest_cat.log 2>&1; tail -5 gpurun_out/r04_pytest_cat.log
for c in 1 0 1 0; do MAGMA_PREFILL_CAT=$c timeout 300 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline --fp8 off --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'cat': $c, 'tokens_per_s': d['value'], 'ms_per_call': d['ms_per_step'], 'token_step_ms': d['roofline']['token_step']['ms']}))" >> gpurun_out/r04_prefill_cat_ab.jsonl; done
cat gpurun_out/r04_prefill_cat_ab.jsonl
