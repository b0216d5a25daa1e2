#!/bin/bash
# end-of-round validation: full GPU test suite, smoke, default bench line, train.py smoke, kernel-trace stats of the bench
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/final_pytest.log 2>&1; tail -3 gpurun_out/final_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
timeout 600 python bench.py > gpurun_out/final_bench.log 2>&1; tail -1 gpurun_out/final_bench.log | cut -c1-400
timeout 400 python train.py --config MAGMA_v1 --synthetic_steps 2 --micro_batch 8 --grad_accum 1 > gpurun_out/final_train_py.log 2>&1; tail -3 gpurun_out/final_train_py.log
cd /tmp; rm -rf $ROOT/gpurun_out/final_trace
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/final_trace -o t -- python $ROOT/bench.py --steps 2 --warmup 1 --train-steps 1 --no-cpu-baseline > $ROOT/gpurun_out/final_trace.log 2>&1
python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/final_trace > $ROOT/gpurun_out/final_trace_summary.txt 2>&1
tail -1 $ROOT/gpurun_out/final_trace.log | cut -c1-300
cp $(find $ROOT/gpurun_out/final_trace -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/final_kernel_stats.csv 2>/dev/null
find $ROOT/gpurun_out/final_trace -name "*.csv" -size +2M -delete
