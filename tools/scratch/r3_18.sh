#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "test_conv133_fwd_bwd" > gpurun_out/r3_tests18.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests18.log | tail -4
python tools/kbench.py L0_32x32d L1_64x64d 2>&1 | grep -E "fwd|dgrad"
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu -s -k "full_size_128_training" > gpurun_out/r3_tests18b.log 2>&1; grep -E "passed|failed|Error|assert|dense-kernel vs" gpurun_out/r3_tests18b.log | tail -6
