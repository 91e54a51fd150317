#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -1
for k in 0 1 0 1; do
echo "== spread $k"
E2E_CONV_SPREAD=$k timeout 120 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "fwd\|dgrad"
done
