#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/phase_timing.py 2>&1 | tail -2 | tee gpurun_out/m_phase.log
cd /tmp
for ph in prefill encoder; do
  rm -rf $ROOT/gpurun_out/tr_$ph
  PHASE=$ph timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tr_$ph -o t -- python $ROOT/tools/prefill_prof.py > $ROOT/gpurun_out/m_$ph.log 2>&1
  python $ROOT/tools/trace_by_grid.py $ROOT/gpurun_out/tr_$ph cast_f32_bf16 | tee $ROOT/gpurun_out/m_${ph}_by_grid.txt
  find $ROOT/gpurun_out/tr_$ph -name "*.csv" -size +2M -delete
done
