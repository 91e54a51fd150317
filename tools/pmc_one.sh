#!/bin/bash
# usage (GPU box, repo root): tools/pmc_one.sh <outdir> <kbench args...>   -- FETCH_SIZE / WRITE_SIZE of single kernels
set -u
OUT=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/$OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_fetch -- python3 $R/tools/kbench.py "$@" > /dev/null 2> $R/gpurun_out/$OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$OUT/pmc_write -- python3 $R/tools/kbench.py "$@" > /dev/null 2> $R/gpurun_out/$OUT/write.err
cd $R
python3 tools/traffic_summary.py gpurun_out/$OUT
