#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests19.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests19.log | tail -6
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --op-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['roofline']['fma']); print(d['op_ms_per_step'])"
python tools/parity_report.py --out gpurun_out/r03_parity.json --tag r3_final_dense_bf3 --net128 > gpurun_out/r3_parity19.log 2>&1; tail -26 gpurun_out/r3_parity19.log | grep -v Total
