# round 6: the same counters as tools/scratch/r05_mm_pmc.sh on the final K1m (64 -> 32 and 32 -> 32 dense @128^3 x 2)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=r06_mm_pmc
mkdir -p $R/gpurun_out/$OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$OUT/pass$i -- python3 $R/tools/kbench.py L0_64x32 L0_32x32d > /dev/null 2> $R/gpurun_out/$OUT/pass$i.err
done
cd $R
python3 tools/pmc_summary.py gpurun_out/$OUT > gpurun_out/$OUT/summary.txt 2>&1
grep -A60 "conv133_mm_kernel" gpurun_out/$OUT/summary.txt | head -150
