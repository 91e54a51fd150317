# round 5: conv133_mm_kernel with parts switched off (timing only): where do the staging wave's cycles go?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_mm; mkdir -p $O
for d in ${MM_DIAGS:-stamps d8 d24 d10}; do
  echo "== build $d (MM_DIAG: 1 no matrix instructions, 2 all plane requests hit one line, 4 no conversion arithmetic)"
  E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_$d.so E2E_MM_STAMPS=1 python tools/kbench.py L0_64x32 2>&1 | grep "conv133_mm mode\|fwd\|dgrad" | awk '/conv133_mm/{k=$3 $4; last[k]=$0; next} {print} END{for (k in last) print last[k]}'
done > $O/diag.txt 2>&1
cat $O/diag.txt
