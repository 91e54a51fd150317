cd $GRAFT_REPO_ROOT
export E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_dbg.so
for d in 0 1 2 3 4 5 6 7; do echo "dbg $d"; E2E_CONV_DBG=$d python tools/kbench.py L0_64x32 2>&1 | grep "plan.*fwd\|plan.*dgrad"; done
