#!/bin/bash
# usage (GPU box, repo root): tools/trace_only.sh <tag>   -- rocprofv3 kernel trace + stats of 3 bench steps
TAG=$1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/bench_trace.json 2> $R/gpurun_out/$TAG/trace.err
