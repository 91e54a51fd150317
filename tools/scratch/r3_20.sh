#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do echo "== DG16=$v"; E2E_CONV_DG16=$v python tools/kbench.py L0_64x32 L1_160x64 2>&1 | grep -E "dgrad"; done
E2E_CONV_DG16=1 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -m gpu -k "conv133" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
