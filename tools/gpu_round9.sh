#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for ph in prefill encoder; do
  rm -rf $ROOT/gpurun_out/tr_$ph
  PHASE=$ph ITERS=4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tr_$ph -o t -- python $ROOT/tools/prefill_prof.py > $ROOT/gpurun_out/tr_$ph.log 2>&1
  python $ROOT/tools/trace_summary.py $ROOT/gpurun_out/tr_$ph > $ROOT/gpurun_out/tr_${ph}_summary.txt 2>&1
  find $ROOT/gpurun_out/tr_$ph -name "*.csv" -size +8M -delete
done
