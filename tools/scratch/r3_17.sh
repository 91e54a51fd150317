#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu -s > gpurun_out/r3_tests17.log 2>&1; grep -E "passed|failed|Error|assert|dense-kernel vs" gpurun_out/r3_tests17.log | tail -6
