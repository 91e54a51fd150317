#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_augment.py -x -q -m gpu > gpurun_out/r3_tests10.log 2>&1; grep -E "passed|failed|Error|assert|error" gpurun_out/r3_tests10.log | tail -12
python tools/scratch/aug_bench.py 2>&1 | tail -4
timeout 900 python -m pytest tests -x -q -m gpu -k "sharded or predict_3d or rccl" > gpurun_out/r3_tests10b.log 2>&1; grep -E "passed|failed|Error|assert|error" gpurun_out/r3_tests10b.log | tail -6
E2E_FORCE_DIST=1 python - <<'PY' 2>&1 | tail -3
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
rec = bench.sliding_window_record(torch.device("cuda", 0), 0, 1)
print(json.dumps(rec))
PY
