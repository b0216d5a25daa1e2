#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sg in 0 0.5 0.25; do
  echo "stagger $sg"; MAGMA_G256_STAGGER=$sg timeout 600 python tools/kbench.py fp8tile 2>&1 | grep fp8tile | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tag'], 'bf16_256 %.3f ms %.0f TF | fp8_256 %.3f ms %.0f TF' % (d['bf16_256_ms'], d['bf16_256_tflops'], d['fp8_256_ms'], d['fp8_256_tflops']))"
done > gpurun_out/par.txt 2>&1
cat gpurun_out/par.txt
