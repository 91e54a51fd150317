# transposed-conv weight gradient: 64-voxel tiles, one workgroup per CU (E2E_CT_TPX32=0) against 32-voxel tiles, two per CU
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do echo "E2E_CT_TPX32=$v"; E2E_CT_TPX32=$v python tools/kbench.py convt 2>&1 | grep wgrad; done; done
