# v5: instructions dealt per matrix-instruction gap (-DWG5_FILL=n; default 5)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in 3 4 5 6 8; do
  lib=$PWD/e2enet_medical_amd/csrc/libe2e_hip_f$f.so; [ $f = 5 ] && lib=$PWD/e2enet_medical_amd/csrc/libe2e_hip.so
  echo "WG5_FILL=$f"; E2E_LIB_PATH=$lib python tools/kbench.py L0_64x32 L1_160x64 2>&1 | grep wgrad
done; done
