# v5 with parts of the stream left out at compile time (-DWG5_DIAG=n; timing only, results wrong): where do the cycles go?
cd $GRAFT_REPO_ROOT
for tgt in 256 64; do
for d in 0 1 2 3 4 7; do
  lib=$PWD/e2enet_medical_amd/csrc/libe2e_hip_g$d.so; [ $d = 0 ] && lib=$PWD/e2enet_medical_amd/csrc/libe2e_hip.so
  echo "WG5_DIAG=$d workgroups=$tgt"; E2E_LIB_PATH=$lib E2E_WG_BF3=5 E2E_WG_V3_TARGET=$tgt python tools/kbench.py L0_64x32 2>&1 | grep wgrad
done; done
