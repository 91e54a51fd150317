#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -n "passed\|failed\|FAILED\|Error" > gpurun_out/all.log
E2E_BENCH_ALL_LAUNCHES=1 python bench.py --steps 10 --warmup 3 --no-extras --op-profile 2>&1 | tail -1 > gpurun_out/lane_bench.json
