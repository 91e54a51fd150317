#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -k "config1 or config5 or whole_net_128 or trainer" -s > gpurun_out/r3_tests5.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/r3_tests5.log | grep -E "grad noise|passed|failed|Error|assert" | tail -20
python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench5.json 2> gpurun_out/r3_bench5.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench5.json')); print(d['ms_per_step'], d['roofline']['frac'], d.get('parity'), d['cpu_baseline'])"
python tools/scratch/node_err.py amos 2>&1 | grep node
