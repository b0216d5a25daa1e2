cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest -q -x -m gpu tests/test_kernels_gpu.py -k "split_k" 2>&1 | tail -2
rm -f gpurun_out/r04_prefill_split256_ab.jsonl
for k in 1 0 1 0; do MAGMA_G256_SPLITK=$k timeout 300 python bench.py --steps 5 --warmup 2 --train-steps 0 --no-cpu-baseline --fp8 off --no-variants 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'MAGMA_G256_SPLITK': '$k', 'tokens_per_s': d['value'], 'ms_per_call': d['ms_per_step']}))" >> gpurun_out/r04_prefill_split256_ab.jsonl; done
cat gpurun_out/r04_prefill_split256_ab.jsonl
