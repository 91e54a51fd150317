# A/B of the weight-gradient variants (E2E_WG_BF3 = 2: v2 twelve waves, 4: v4 matrix + staging waves, 5: v5 one stream per SIMD)
cd $GRAFT_REPO_ROOT
for v in ${WG_VARIANTS:-2 4 5 2 4 5}; do echo "E2E_WG_BF3=$v"; E2E_WG_BF3=$v python tools/kbench.py L0_64x32 L1_160x64 L2_320x128 L0_32x32d 2>&1 | grep wgrad; done
