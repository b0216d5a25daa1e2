#!/bin/bash
# rocprofv3 kernel trace (+stats) of the bench command, then two PMC passes.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_trace -o bench -- $CMD > $ROOT/gpurun_out/prof_trace.log 2>&1
tail -2 $ROOT/gpurun_out/prof_trace.log
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $ROOT/gpurun_out/prof_fetch -o bench -- $CMD > $ROOT/gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $ROOT/gpurun_out/prof_write -o bench -- $CMD > $ROOT/gpurun_out/prof_write.log 2>&1
cd $ROOT
find gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30
# keep only the small summaries + a compressed kernel trace (scratch dir is capped at 64 MiB)
for d in prof_trace prof_fetch prof_write; do
  find gpurun_out/$d -name "*.db" -delete 2>/dev/null
  find gpurun_out/$d -name "*_kernel_trace.csv" -size +20M -exec sh -c 'head -200000 "$1" > "$1.head"; rm "$1"' _ {} \;
done
python tools/prof_summary.py gpurun_out > gpurun_out/prof_summary.txt 2>&1
cat gpurun_out/prof_summary.txt | head -60
