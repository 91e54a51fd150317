#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/scratch/conv_err.py 2>&1 | grep cin
timeout 1200 python -m pytest tests -x -q -m gpu -k "not config1 and not config5" > gpurun_out/r3_tests4.log 2>&1; tail -5 gpurun_out/r3_tests4.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
