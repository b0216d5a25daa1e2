#!/bin/bash
# End-of-round evidence on one MI355X (ROUND=rNN, default r04; SKIP_PYTEST / SKIP_ATTN_PMC / SKIP_GEMM_PMC = 1 leave parts out): full GPU test suite, smoke, the default bench line, train.py
# smoke, kernel-trace stats of the bench, the PMC FETCH_SIZE pass behind roofline.traffic, and per-(kernel, grid) traces of the
# decode step, the prefill and one training step.  Everything lands in gpurun_out/; copy what is to be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp; R=${ROUND:-r04}
[ "${SKIP_PYTEST:-0}" = "1" ] || { timeout 1500 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/${R}_final_pytest_gpu.log 2>&1; tail -3 gpurun_out/${R}_final_pytest_gpu.log; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_final_smoke.log 2>&1; tail -2 gpurun_out/${R}_final_smoke.log
cd /tmp; rm -rf $ROOT/gpurun_out/pmc_dec
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_dec -o t -- python $ROOT/tools/pmc_decode_sweep.py > $ROOT/gpurun_out/pmc_dec.log 2>&1
python $ROOT/tools/pmc_table.py $ROOT/gpurun_out/pmc_dec $ROOT/gpurun_out/${R}_decode_gemv_fetch_table.json | tail -3
find $ROOT/gpurun_out/pmc_dec -name "*.csv" -size +4M -delete
mkdir -p $ROOT/profiles; cp $ROOT/gpurun_out/${R}_decode_gemv_fetch_table.json $ROOT/profiles/ 2>/dev/null   # bench.py reads the newest table
cd $ROOT
timeout 600 python bench.py > gpurun_out/${R}_final_bench.json 2> gpurun_out/${R}_final_bench.err; tail -c 600 gpurun_out/${R}_final_bench.json
timeout 400 python train.py --config MAGMA_v1 --synthetic_steps 2 --micro_batch 8 --grad_accum 1 > gpurun_out/${R}_final_train_py.log 2>&1; tail -3 gpurun_out/${R}_final_train_py.log
cd /tmp; rm -rf $ROOT/gpurun_out/final_trace
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/final_trace -o t -- python $ROOT/bench.py --steps 2 --warmup 1 --train-steps 1 --no-cpu-baseline --no-variants > $ROOT/gpurun_out/${R}_final_bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/final_trace > $ROOT/gpurun_out/${R}_final_bench_kernel_trace_summary.txt 2>&1
cp $(find $ROOT/gpurun_out/final_trace -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${R}_final_bench_kernel_stats.csv 2>/dev/null
find $ROOT/gpurun_out/final_trace -name "*.csv" -size +2M -delete
# decode step / prefill / training step per (kernel, grid)
rm -rf $ROOT/gpurun_out/decode_trace; TRACE_MARK=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/decode_trace -o t -- python $ROOT/tools/decode_step_bench.py > /dev/null 2>&1
python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/decode_trace cast_f32_bf16 > $ROOT/gpurun_out/${R}_decode_trace_by_grid.txt 2>&1
rm -rf $ROOT/gpurun_out/prefill_trace; PHASE=prefill timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prefill_trace -o t -- python $ROOT/tools/prefill_prof.py > /dev/null 2>&1
python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/prefill_trace cast_f32_bf16 > $ROOT/gpurun_out/${R}_prefill_trace_by_grid.txt 2>&1
rm -rf $ROOT/gpurun_out/train_trace; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/train_trace -o t -- python $ROOT/tools/train_trace.py > /dev/null 2>&1
TOP=45 python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/train_trace advance_pos > $ROOT/gpurun_out/${R}_train_step_kernel_trace_summary.txt 2>&1
TOP=70 python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/train_trace advance_pos > $ROOT/gpurun_out/${R}_train_trace_by_grid.txt 2>&1 || true
for d in decode_trace prefill_trace train_trace; do find $ROOT/gpurun_out/$d -name "*.csv" -size +2M -delete; done
head -8 $ROOT/gpurun_out/${R}_decode_trace_by_grid.txt; head -6 $ROOT/gpurun_out/${R}_prefill_trace_by_grid.txt; head -12 $ROOT/gpurun_out/${R}_train_step_kernel_trace_summary.txt
# attention PMC summary of the round (six counter passes over tools/attn_bench.py)
[ "${SKIP_ATTN_PMC:-0}" = "1" ] || { bash $ROOT/tools/gpu_pmc_attn.sh > /dev/null 2>&1; cp $ROOT/gpurun_out/pmc_attn_summary.txt $ROOT/gpurun_out/${R}_attention_pmc_summary.txt 2>/dev/null; tail -8 $ROOT/gpurun_out/${R}_attention_pmc_summary.txt; }
# MFMA-pipe busy / instruction mix of the 256x256 GEMM next to hipBLASLt's kernel on the training shapes (the kernel behind roofline.train)
[ "${SKIP_GEMM_PMC:-0}" = "1" ] || { bash $ROOT/tools/gpu_pmc_gemm_util.sh > /dev/null 2>&1; cp $ROOT/gpurun_out/pmc_gemm_util_summary.txt $ROOT/gpurun_out/${R}_gemm256_pmc_summary.txt 2>/dev/null; tail -12 $ROOT/gpurun_out/${R}_gemm256_pmc_summary.txt; }
