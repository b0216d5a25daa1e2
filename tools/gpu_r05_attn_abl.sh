#!/bin/bash
# round 5: timing ablations of attn_bwd_dkdv32_kernel (ablation library; WRONG results): each arm's price is the time it removes
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MAGMA_HIP_LIB=$(pwd)/magma_amd/libmagma_hip_abl.so
rm -f gpurun_out/r05_attn_bwd32_abl.jsonl
for abl in 0 1 2 3 4 5 0; do
  MAGMA_ATTN_BWD32_ABL=$abl AB=16 ABWD=${ABWD:-2} timeout 200 python tools/attn_bench.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'abl': $abl, **{k: v for k, v in d.items() if k.startswith('bwd_ms')}}))" >> gpurun_out/r05_attn_bwd32_abl.jsonl
done
cat gpurun_out/r05_attn_bwd32_abl.jsonl
