cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_backward_kernels_gpu.py tests/test_variants_gpu.py tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py -q -m gpu -k "transpose_colsum or options or gradients or train" 2>&1 | tail -5
timeout 200 python tools/adapter_gemm_bench.py 2>/dev/null | tee gpurun_out/r04_adapter_gemm_tiles.jsonl
