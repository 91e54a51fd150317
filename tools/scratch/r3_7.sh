#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests7.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests7.log | tail -8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --op-profile > gpurun_out/r3_bench7.json 2> gpurun_out/r3_bench7.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench7.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline_secondary']); print(d['op_ms_per_step'])"
