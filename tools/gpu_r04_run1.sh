#!/bin/bash
# round 4, first GPU pass: full GPU suite, decode A/B (three- vs four-launch block), default bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r04_pytest_gpu_first.log 2>&1; tail -25 gpurun_out/r04_pytest_gpu_first.log
for f in 0 1; do MAGMA_DECODE_FOLD=$f timeout 300 python tools/decode_step_bench.py 2>/dev/null | head -1 >> gpurun_out/r04_decode_fold_ab.jsonl; done
cat gpurun_out/r04_decode_fold_ab.jsonl
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04_bench_first.json 2> gpurun_out/r04_bench_first.err; tail -c 1500 gpurun_out/r04_bench_first.json; tail -3 gpurun_out/r04_bench_first.err
