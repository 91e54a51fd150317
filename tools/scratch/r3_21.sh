#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "graph" > gpurun_out/r3_tests21.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests21.log | tail -6
echo "== eager"; E2E_GRAPHS=0 python tools/scratch/small_bench.py 2>&1 | grep wall
echo "== graphs (auto)"; python tools/scratch/small_bench.py 2>&1 | grep wall
