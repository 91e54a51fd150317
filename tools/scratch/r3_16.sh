#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "test_conv133_fwd_bwd" > gpurun_out/r3_tests16.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests16.log | tail -4
python tools/kbench.py L0_32x32d L1_64x64d L0_64x32_d05 2>&1 | grep -E "fwd|dgrad"
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests16b.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests16b.log | tail -6
