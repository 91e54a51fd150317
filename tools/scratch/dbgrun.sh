#!/bin/bash
cd $GRAFT_REPO_ROOT
cp e2enet_medical_amd/csrc/libe2e_hip.so /tmp/lib_new.so
for rep in 1 2 3; do
for v in old new; do
if [ $v = old ]; then cp tools/scratch/lib_old.so e2enet_medical_amd/csrc/libe2e_hip.so; else cp /tmp/lib_new.so e2enet_medical_amd/csrc/libe2e_hip.so; fi
echo "== $v"
timeout 120 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "fwd\|dgrad"
done
done
cp /tmp/lib_new.so e2enet_medical_amd/csrc/libe2e_hip.so
