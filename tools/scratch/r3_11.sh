#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests11.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests11.log | tail -8
E2E_FORCE_DIST=1 python - > gpurun_out/r3_sw_forced.json 2> gpurun_out/r3_sw_forced.err <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
import bench
rec = bench.sliding_window_record(torch.device("cuda", 0), 0, 1)
sys.stdout.write(json.dumps(rec) + "\n"); sys.stdout.flush()
dist.destroy_process_group()
PY
grep workload gpurun_out/r3_sw_forced.json | cut -c1-900
