# timing-only sweep of the weight-gradient kernel (v2 schedule) with phases switched off (diagnostic build)
cd $GRAFT_REPO_ROOT
export E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_dbg.so
for d in 0 1 2 3 4 5 7 8 9 11 12 15; do echo "dbg $d"; E2E_WG_BF3=2 E2E_WG_DBG=$d python tools/kbench.py L0_64x32 2>&1 | grep "wgrad"; done
