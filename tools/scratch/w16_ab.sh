# weight gradient of 16-wide planes: fp32-MFMA v2 kernel (E2E_WG_W16=0) against the bf16x3 kernel on 8 x 16 tiles
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do echo "E2E_WG_W16=$v"; E2E_WG_W16=$v python tools/kbench.py L3_640x256 2>&1 | grep wgrad; done; done
