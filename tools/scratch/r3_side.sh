#!/bin/bash
mkdir -p gpurun_out
for v in "0 0" "1 0" "1 1"; do
  set -- $v
  echo "== wgrad_stream=$1 lanes=$2" >> gpurun_out/side.log
  E2E_WGRAD_STREAM=$1 E2E_LANES=$2 python bench.py --steps 10 --warmup 3 --no-extras 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/side.log 2>&1
done
E2E_LANE_LIGHT_DIV=8 python bench.py --steps 10 --warmup 3 --no-extras 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('div8', d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/side.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | grep -n "passed\|failed\|FAILED\|Error" >> gpurun_out/side.log
