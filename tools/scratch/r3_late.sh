#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/late.log
for v in 0 1 0 1; do
  E2E_WGRAD_LATE=$v python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('late=$v', d['ms_per_step'])" >> gpurun_out/late.log 2>&1
done
