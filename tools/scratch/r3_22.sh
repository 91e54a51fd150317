#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests22.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests22.log | tail -6
