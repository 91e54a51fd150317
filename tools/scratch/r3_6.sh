#!/bin/bash
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/bf3_numerics tools/scratch/bf3_numerics.hip && /tmp/bf3_numerics
echo "== wgrad op tests (bf3 path)"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -m gpu -k "conv133" > gpurun_out/r3_tests6.log 2>&1; grep -E "passed|failed|Error|assert|wgrad" gpurun_out/r3_tests6.log | tail -15
echo "== kbench bf3"
python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 2>&1 | grep wgrad
echo "== kbench v3"
E2E_WG_BF3=0 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 2>&1 | grep wgrad
python tools/scratch/conv_err.py 2>&1 | grep cin | tail -4
