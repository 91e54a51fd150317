#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv133" 2>&1 | tail -3 > gpurun_out/diag.log
python tools/kbench.py L0_32x32d L0_64x32_d05 2>&1 | grep -v amdgpu.ids >> gpurun_out/diag.log
E2E_LIB_PATH=$GRAFT_REPO_ROOT/e2enet_medical_amd/csrc/libe2e_d4.so python tools/kbench.py L0_32x32d 2>&1 | grep WG | tail -4 >> gpurun_out/diag.log
