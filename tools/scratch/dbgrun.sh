#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -x -q 2>&1 | tail -2
for k in 1 0; do
echo "== ksplit knob $k (1 = off, 0... -1 default)"
done
E2E_CONV_KSPLIT=1 timeout 200 python tools/scratch/small_bench.py 2>&1 | grep wall
echo "== default (split-K on)"
timeout 200 python tools/scratch/small_bench.py 2>&1 | grep wall
E2E_CONV_KSPLIT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('off', d['ms_per_step'], d['roofline']['ms_per_step'])"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('on ', d['ms_per_step'], d['roofline']['ms_per_step'])"
