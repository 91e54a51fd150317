#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu -k "conv133 or whole_net or config or tiny or net64 or sparse or nodff" > gpurun_out/r3_tests15.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_tests15.log | tail -8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --op-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step']); print(d['op_ms_per_step'])"
export KB_ONLY=fwd
bash tools/pmc_mfma.sh pmc_dense L0_32x32d 2>&1 | grep -A17 "dense_kernel" | head -40
