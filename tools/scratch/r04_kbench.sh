# round-4 kernels against the round-3 paths on the same box (kbench: single launches, 10 back to back)
cd $GRAFT_REPO_ROOT
echo "# round 4 (default paths: planned sparse walk, weight gradient v5)"
python tools/kbench.py L0_64x32 L1_160x64 L2_320x128 L0_32x32d L1_s2_32x64d 2>&1 | grep -v amdgpu
echo "# round-3 paths on this box (E2E_CONV_SPARSE2=0 KB_OLD=1 E2E_WG_BF3=2)"
E2E_CONV_SPARSE2=0 KB_OLD=1 E2E_WG_BF3=2 python tools/kbench.py L0_64x32 L1_160x64 L2_320x128 L0_32x32d L1_s2_32x64d 2>&1 | grep -v amdgpu
