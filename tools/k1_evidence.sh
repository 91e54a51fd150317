#!/bin/bash
# usage (GPU box, repo root): bash tools/k1_evidence.sh <tag>
# Evidence for the DSFF-masked conv walk (conv133_kernel<0|1,...,16,32,...>) on 64->32 @128^3 x 2:
#   1) time against density (KB_DENSITY 0.02 .. 0.5), shipping build
#   2) per-phase s_memtime table of the same kernel (diagnostic build: make BUILD=build_dbg LIB=libe2e_hip_dbg.so DEFS=-DE2E_CONV_DEBUG,
#      E2E_CONV_DBG=8), forward and data gradient
#   3) SQ counters, separate --pmc passes (tools/pmc_lds.sh)
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
for d in 0.02 0.1 0.2 0.3 0.5; do
  echo "density $d" >> $O/density.log
  KB_DENSITY=$d KB_NO_DENSE=1 timeout 300 python tools/kbench.py L0_64x32 2>&1 | grep -v amdgpu.ids >> $O/density.log
done
E2E_LIB_PATH=$R/e2enet_medical_amd/csrc/libe2e_hip_dbg.so E2E_CONV_DBG=8 timeout 300 python tools/kbench.py L0_64x32 L1_160x64 > $O/phases.log 2>&1
bash tools/pmc_lds.sh $TAG/pmc_lds L0_64x32 > $O/pmc_lds.txt 2>&1
bash tools/pmc.sh $TAG/pmc L0_64x32 > $O/pmc.txt 2>&1
tail -30 $O/density.log
grep "conv133 MODE" $O/phases.log | sort | uniq -c | sort -rn | head -20
