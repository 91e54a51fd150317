# staging-wave priority of the weight-gradient kernel v4 (libraries built with -DE2E_WG4_STAGE_PRIO=n)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for p in 0 1 2 3; do
  lib=$PWD/e2enet_medical_amd/csrc/libe2e_hip_p$p.so; [ $p = 2 ] && lib=$PWD/e2enet_medical_amd/csrc/libe2e_hip.so
  echo "prio $p"; E2E_LIB_PATH=$lib python tools/kbench.py L0_64x32 L1_160x64 L2_320x128 2>&1 | grep wgrad
done; done
