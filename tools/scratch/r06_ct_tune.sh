cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_ct; mkdir -p $O
{
for f in 512 768 1024; do
  echo "--- E2E_CT_TUNE_FWD=$f E2E_CT_TUNE_DG=$f"
  E2E_CT_TUNE_FWD=$f E2E_CT_TUNE_DG=$f timeout 300 python tools/kbench.py up_L0_64x32 up_L1_128x64 2>&1 | grep -v "amdgpu.ids\|unknown E2E\|warnings.warn"
done
echo "--- again 512 / 768"
for f in 512 768; do
  echo "--- E2E_CT_TUNE_FWD=$f"
  E2E_CT_TUNE_FWD=$f timeout 300 python tools/kbench.py up_L0_64x32 2>&1 | grep -v "amdgpu.ids\|unknown E2E\|warnings.warn"
done
} > $O/out.txt 2>&1
cat $O/out.txt
