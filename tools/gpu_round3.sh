#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_backward_kernels_gpu.py -m gpu -q --timeout 300 2>&1 | tail -60 > gpurun_out/pytest_bwd.log
grep -E "passed|failed|error" gpurun_out/pytest_bwd.log
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q --timeout 600 -s 2>&1 | tail -120 > gpurun_out/pytest_train.log
grep -E "passed|failed|error|worst" gpurun_out/pytest_train.log | cut -c1-600
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 2>&1 | tail -30 > gpurun_out/pytest_fwd.log
grep -E "passed|failed|error" gpurun_out/pytest_fwd.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_trace -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_trace.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof_fetch -o sk -- python $ROOT/tools/pmc_skinny.py > $ROOT/gpurun_out/prof_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_write -o sk -- python $ROOT/tools/pmc_skinny.py > $ROOT/gpurun_out/prof_write.log 2>&1
cd $ROOT
find gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30
for d in prof_trace prof_fetch prof_write; do
  find gpurun_out/$d -name "*kernel_trace.csv" -size +8M -exec sh -c 'head -60000 "$1" > "$1.head"; rm "$1"' _ {} \;
done
python tools/prof_summary.py gpurun_out > gpurun_out/prof_summary.txt 2>&1
head -70 gpurun_out/prof_summary.txt | cut -c1-220
