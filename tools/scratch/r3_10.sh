#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_augment.py -x -q -m gpu > gpurun_out/r3_tests10.log 2>&1; grep -E "passed|failed|Error|assert|error" gpurun_out/r3_tests10.log | tail -12
python tools/scratch/aug_bench.py 2>&1 | tail -4
