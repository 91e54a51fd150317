#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <outdir> <kbench args...>
# separate rocprofv3 --pmc passes (no trace domains), csv output under gpurun_out/<outdir>/passN
set -u
OUT=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/$OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/pass$i -- python3 $R/tools/kbench.py "$@" > /dev/null 2> $R/gpurun_out/$OUT/pass$i.err
done
cd $R
python3 tools/pmc_summary.py gpurun_out/$OUT
