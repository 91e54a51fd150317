#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -m gpu -k "conv133" > gpurun_out/r3_tests12.log 2>&1; grep -E "passed|failed|Error|assert|wgrad" gpurun_out/r3_tests12.log | tail -8
for v in 2 1 2 1; do echo "== bf3 variant $v"; E2E_WG_BF3=$v python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 2>&1 | grep wgrad; done
for g in 64 0 32 128; do echo "== in_bwd group MB $g"; E2E_IN_BWD_GROUP_MB=$g python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --op-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['op_ms_per_step']['in_lrelu_bwd'], d['op_ms_per_step']['conv133_wgrad'])"; done
