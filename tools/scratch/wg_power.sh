# is the weight gradient power-bound?  the same work on 256 / 128 / 64 workgroups (one per CU): per-CU speed against CUs in use
cd $GRAFT_REPO_ROOT
for v in 4 5; do for tgt in 256 128 64; do echo "E2E_WG_BF3=$v workgroups=$tgt"; E2E_WG_BF3=$v E2E_WG_V3_TARGET=$tgt python tools/kbench.py L0_64x32 2>&1 | grep wgrad; done; done
