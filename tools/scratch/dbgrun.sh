#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv133_fwd_bwd" 2>&1 | grep "passed\|failed"
for rep in 1 2; do
for dbg in 64 0; do
echo "== wg dbg $dbg (64 = both wave groups stage at the same k-steps)"
E2E_WG_DBG=$dbg timeout 120 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "wgrad"
done
done
