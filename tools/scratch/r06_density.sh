# verdict r05 weak 12: is the dense matrix-pipe conv (K1m) still ahead of the round-4 sparse walk at final_density 0.05 / 0.1?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_density; mkdir -p $O
{
for r in 1 2; do
  echo "--- K1m (default)"
  timeout 300 python tools/kbench.py L0_64x32_d005 L0_64x32_d01 L0_64x32 L1_160x64_d005 2>&1 | grep -v "amdgpu.ids" | grep "fwd\|dgrad"
  echo "--- E2E_CONV_MM=0 (round-4 sparse walk)"
  E2E_CONV_MM=0 timeout 300 python tools/kbench.py L0_64x32_d005 L0_64x32_d01 L0_64x32 L1_160x64_d005 2>&1 | grep -v "amdgpu.ids\|unknown E2E" | grep "fwd\|dgrad"
done
} > $O/out.txt 2>&1
tail -40 $O/out.txt
