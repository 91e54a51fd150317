#!/bin/bash
cd $GRAFT_REPO_ROOT
for k in 0 1; do
echo "== sched $k"
E2E_WG_SCHED=$k timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or conv133" 2>&1 | tail -1
E2E_WG_SCHED=$k timeout 120 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "wgrad"
done
