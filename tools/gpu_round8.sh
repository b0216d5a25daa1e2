#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_backward_kernels_gpu.py -m gpu -q --timeout 600 2>&1 | tail -30 > gpurun_out/pytest_train2.log
grep -E "passed|failed|error|Error" gpurun_out/pytest_train2.log | head -20
timeout 600 python bench.py --steps 2 --warmup 1 --train-steps 2 --no-cpu-baseline > gpurun_out/bench6.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench6.log') if x.startswith('{')][-1]; d=json.loads(l)
print(d['value'], d['train'])
PY
