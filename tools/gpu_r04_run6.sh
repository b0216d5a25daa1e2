#!/bin/bash
# round 4, sixth GPU pass: split-K form of the 256x256 kernel (test, wgrad shapes in isolation, training step)
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "split_k or gemm256" 2>&1 | tail -4
timeout 120 python tools/wgrad_bench.py 2>/dev/null | tee gpurun_out/r04_wgrad_split256.jsonl
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fp8 off --no-variants --no-train-truncate 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print(json.dumps({'forward_only_ms': t['forward_only']['ms'], 'full_S2048': t['full_S2048'], 'roofline_train': d['roofline'].get('train')}))" | tee gpurun_out/r04_train_after_split256.json
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_fullwidth_train_gpu.py -q -m gpu -x 2>&1 | tail -3
