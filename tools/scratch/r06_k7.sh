# K7 apply pass A/B (instnorm.hip -DIN_BWD_VARIANT=n: 1 two chunks per iteration, 2 nontemporal loads of y), interleaved twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_k7; mkdir -p $O
{
for r in 1 2; do
for v in 0 1 2 3; do
  echo "--- IN_BWD_VARIANT=$v"
  E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_k7v$v.so timeout 300 python tools/scratch/k7_bench.py 2>&1 | grep -v "amdgpu.ids"
done
done
} > $O/out.txt 2>&1
cat $O/out.txt
