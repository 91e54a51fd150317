#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -k "convT" 2>&1 | tail -1
timeout 120 python tools/kbench.py convt 2>&1 | grep "up_"
