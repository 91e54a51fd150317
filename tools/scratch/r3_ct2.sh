#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "convT" 2>&1 | tail -3 > gpurun_out/ct.log
python tools/kbench.py up_L0_64x32 up_L1_128x64 2>&1 | grep -v amdgpu.ids >> gpurun_out/ct.log
E2E_LIB_PATH=$GRAFT_REPO_ROOT/e2enet_medical_amd/csrc/libe2e_ct.so python tools/kbench.py up_L0_64x32 2>&1 | grep WG | tail -3 >> gpurun_out/ct.log
