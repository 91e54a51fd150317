#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "convT" 2>&1 | tail -5 > gpurun_out/ct.log
python tools/kbench.py convt 2>&1 | grep -v amdgpu.ids >> gpurun_out/ct.log
echo "== fp32 mfma (v2/v3)" >> gpurun_out/ct.log
E2E_CT_BF3=0 python tools/kbench.py convt 2>&1 | grep grad >> gpurun_out/ct.log
