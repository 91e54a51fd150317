#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for w in 100000 1024 512 2048; do
echo "== wgs $w"
E2E_CONV_WGS=$w timeout 120 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "fwd\|dgrad"
done
