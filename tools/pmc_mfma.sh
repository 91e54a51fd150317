#!/bin/bash
# usage (GPU box, repo root): tools/pmc_mfma.sh <outdir> <kbench args...>   -- matrix-pipe / issue counters (weight gradient)
set -u
OUT=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/pass$i -- python3 $R/tools/kbench.py "$@" > /dev/null 2> $R/gpurun_out/$OUT/pass$i.err
done
cd $R
python3 tools/pmc_summary.py gpurun_out/$OUT
