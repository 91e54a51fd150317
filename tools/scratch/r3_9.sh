#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -k "nodff or resample or export or predict_cases" > gpurun_out/r3_tests9.log 2>&1; grep -E "passed|failed|Error|assert|error" gpurun_out/r3_tests9.log | tail -12
