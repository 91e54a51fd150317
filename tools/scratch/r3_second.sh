#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp profiles/r03_parity.json gpurun_out/r03_parity.json 2>/dev/null
python tools/parity_report.py --out gpurun_out/r03_parity.json --tag r3_default_twolevel > gpurun_out/r3_parity_new.log 2>&1
tail -22 gpurun_out/r3_parity_new.log
python tools/scratch/node_err.py hippo 2>&1 | tail -25
python tools/scratch/conv_err.py 2>&1 | tail -6
