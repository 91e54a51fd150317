#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/prio.log
for v in "0 0" "-1 0" "0 -1" "-1 -1" "0 0" "-1 0"; do
  set -- $v
  E2E_LANE_PRIORITY=$1 E2E_WGRAD_PRIORITY=$2 python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('lane prio $1 wgrad prio $2', d['ms_per_step'])" >> gpurun_out/prio.log 2>&1
done
