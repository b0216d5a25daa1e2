#!/bin/bash
# FETCH_SIZE of every decode GEMV shape on the full-size model (separate PMC pass, kernel-trace only) -> per-shape table
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
rm -rf $ROOT/gpurun_out/pmc_dec
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_dec -o t -- python $ROOT/tools/pmc_decode_sweep.py > $ROOT/gpurun_out/pmc_dec.log 2>&1
tail -1 $ROOT/gpurun_out/pmc_dec.log
python $ROOT/tools/pmc_table.py $ROOT/gpurun_out/pmc_dec $ROOT/gpurun_out/r02_decode_gemv_fetch_table.json
find $ROOT/gpurun_out/pmc_dec -name "*.csv" -size +4M -delete
