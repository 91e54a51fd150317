#!/bin/bash
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o tools/scratch/walk_bench tools/scratch/walk_bench.hip && tools/scratch/walk_bench 2>&1 | tail -16
echo "== rocm-smi idle"; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk" | head -6
echo "== conv kernel loop"
(timeout 60 python tools/scratch/loop_conv.py > gpurun_out/loop.log 2>&1 &)
sleep 28
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power\|mclk" | tr '\n' ' '; echo; sleep 1; done
sleep 20
tail -3 gpurun_out/loop.log
