#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 0 2 1; do
echo "== OPW8 knob $k"
E2E_CONV_OPW8=$k timeout 120 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "fwd\|dgrad"
done
