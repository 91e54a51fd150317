#!/bin/bash
# usage (GPU box, repo root): tools/prof_bench.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of bench.py  2) two separate --pmc passes (FETCH_SIZE / WRITE_SIZE)
TAG=$1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/$TAG/bench_trace.json 2> $R/gpurun_out/$TAG/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/$TAG/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/$TAG/pmc_write.err
cd $R
python3 tools/traffic_summary.py gpurun_out/$TAG
