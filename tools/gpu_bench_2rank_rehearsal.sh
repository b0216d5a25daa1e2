#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on a ONE-GPU box: bench.py launches its two ranks itself (both on device 0, gloo instead of RCCL,
# which refuses two ranks per device), 2 LM layers, tiny training batch.  Checks that every rank issues the same collectives in the
# same order (no hang) and that rank 0 prints one JSON line with n_gpus = 2 and the data-parallel fields; the numbers mean nothing.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
MAGMA_BENCH_BACKEND=gloo MAGMA_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --layers 2 --train-steps 1 --train-warmup 1 \
  --train-batch 2 --cpu-seconds 8 > gpurun_out/bench_2rank_rehearsal.json 2> gpurun_out/bench_2rank_rehearsal.err
echo "rc=$?"; tail -c 1800 gpurun_out/bench_2rank_rehearsal.json; tail -5 gpurun_out/bench_2rank_rehearsal.err
# the short-lease record: training leg only
MAGMA_BENCH_BACKEND=gloo MAGMA_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --train-only --layers 2 --train-steps 1 --train-warmup 1 \
  --train-batch 2 --cpu-seconds 8 --fp8 off --no-train-truncate > gpurun_out/bench_2rank_train_only.json 2> gpurun_out/bench_2rank_train_only.err
echo "rc=$?"; tail -c 1500 gpurun_out/bench_2rank_train_only.json; tail -3 gpurun_out/bench_2rank_train_only.err
