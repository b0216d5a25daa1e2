cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 200 python tools/adapter_gemm_bench.py 2>/dev/null | tee gpurun_out/r04_adapter_gemm_tiles_pipelined_walk.jsonl
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_backward_kernels_gpu.py tests/test_fp8_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --train-steps 4 --no-cpu-baseline --no-variants --no-train-truncate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['train']
print(json.dumps({'forward_only_ms': t['forward_only']['ms'], 'train_ms': t['full_S2048']['ms_per_step'], 'train_fp8_ms': t['full_S2048_fp8']['ms_per_step'], 'gemm_roofline': d['roofline']['train']['achieved']}))"
