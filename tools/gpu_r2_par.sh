#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --fp8 off --train-steps 0 2>&1 | tail -1 > gpurun_out/par.txt
python -c "
import json; d=json.loads(open('gpurun_out/par.txt').read()); print(d['value'], d['ms_per_step'], d['generate_from_host'], d['generate_sampled'])"
