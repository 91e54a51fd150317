# round 5 before / after on one box: the round-4 kernels (E2E_CONV_MM=0 E2E_WG_H2=0: load-balanced sparse walk, bf16x3 dense conv,
# bf16x3 weight gradient) against the round-5 defaults (fp16 two-piece persistent matrix-pipe conv K1m, fp16 two-piece weight gradient);
# interleaved twice.  -> gpurun_out/r05_kbench/kbench.txt (summary: profiles/r05_kbench.txt)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_kbench; mkdir -p $O
CASES="L0_64x32 L0_64x32_d05 L0_32x32d L1_160x64 L1_64x64d L2_320x128"
{
for rep in 1 2; do
  echo "# round-4 kernels on this box (E2E_CONV_MM=0 E2E_WG_H2=0), pass $rep"
  E2E_CONV_MM=0 E2E_WG_H2=0 KB_ITERS=30 python tools/kbench.py $CASES 2>&1 | grep "fwd\|dgrad\|wgrad\|pack"
  echo "# round-5 defaults, pass $rep"
  KB_ITERS=30 python tools/kbench.py $CASES 2>&1 | grep "fwd\|dgrad\|wgrad\|pack"
done
} > $O/kbench.txt 2>&1
tail -40 $O/kbench.txt
