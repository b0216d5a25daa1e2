#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 -k "skinny or decode or model or generate or prefill or smoke" 2>&1 | tail -40 > gpurun_out/pytest_dec3.log
grep -E "passed|failed|error|Error" gpurun_out/pytest_dec3.log | head
rm -f gpurun_out/kbench.jsonl
timeout 300 python tools/kbench.py skinny > gpurun_out/kbench2.log 2>&1
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/kbench.jsonl') if l.strip()]
best={}
for r in rows:
    if r.get('kind')!='skinny' or 'ms' not in r: continue
    k=r['tag']; best.setdefault(k, []).append(r)
for k,v in best.items():
    v.sort(key=lambda r:r['ms'])
    print(k, [(r['nt'],r['waves'],r['kc'],round(r['ms']*1e3,1),int(r['gbps'])) for r in v[:4]])
PY
timeout 600 python bench.py --steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline > gpurun_out/bench5.log 2>&1
tail -c 1800 gpurun_out/bench5.log
MAGMA_DECODE_STREAMS=1 timeout 600 python bench.py --steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline > gpurun_out/bench5_1stream.log 2>&1
tail -c 700 gpurun_out/bench5_1stream.log
