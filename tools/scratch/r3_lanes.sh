#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/lanes.log
for dv in "64" "64,4096" "8,64,512" "8,512"; do
  echo "== small E2E_LANE_DIVS=$dv" >> gpurun_out/lanes.log
  E2E_LANE_DIVS=$dv python tools/scratch/small_bench.py 2>&1 | grep "fwd+loss\|64^3  " >> gpurun_out/lanes.log
done
for dv in "64" "64,4096" "64,512" "64,512,4096"; do
  E2E_LANE_DIVS=$dv python bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('128^3 lanes=$dv', d['ms_per_step'])" >> gpurun_out/lanes.log 2>&1
done
python -m pytest tests/test_gpu_net.py -m gpu -x -q -k "two_lane or graph_replay" 2>&1 | tail -2 >> gpurun_out/lanes.log
