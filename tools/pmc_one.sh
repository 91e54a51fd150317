#!/bin/bash
# usage: tools/pmc_one.sh <outdir> "<counters>" <kbench args...>   (one rocprofv3 --pmc pass)
OUT=$1; CTR=$2; shift; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$OUT
cd /tmp
rocprofv3 --pmc $CTR --output-format csv -d $R/gpurun_out/$OUT/pass1 -- python3 $R/tools/kbench.py "$@" > /dev/null 2> $R/gpurun_out/$OUT/pass1.err
cd $R
python3 tools/pmc_summary.py gpurun_out/$OUT
