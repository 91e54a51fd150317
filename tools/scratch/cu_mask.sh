# are the hot kernels power-limited at full occupancy?  the same launches confined to 64 / 128 of the 256 CUs (ROC_GLOBAL_CU_MASK)
cd $GRAFT_REPO_ROOT
M64=0xffffffffffffffff
M128=0xffffffffffffffffffffffffffffffff
echo "all CUs";            python tools/kbench.py L0_64x32 2>&1 | grep "fwd\|dgrad\|wgrad"
echo "128 CUs";  ROC_GLOBAL_CU_MASK=$M128 python tools/kbench.py L0_64x32 2>&1 | grep "fwd\|dgrad\|wgrad"
echo "64 CUs";   ROC_GLOBAL_CU_MASK=$M64  python tools/kbench.py L0_64x32 2>&1 | grep "fwd\|dgrad\|wgrad"
