#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp profiles/r03_parity.json gpurun_out/r03_parity.json 2>/dev/null
python tools/parity_report.py --out gpurun_out/r03_parity.json --tag r3_twolevel_stats64 > gpurun_out/r3_parity_new2.log 2>&1
tail -22 gpurun_out/r3_parity_new2.log
python tools/scratch/node_err.py hippo 2>&1 | tail -22
timeout 900 python -m pytest tests -x -q -m gpu -k "not config1 and not config5" 2>&1 | tail -5
