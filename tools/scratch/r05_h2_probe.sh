# round 5, call 1: fp16x2 numerics probe, fp16x2 weight gradient against the bf16x3 kernel (time and error)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_h2; mkdir -p $O
./tools/scratch/h2_numerics.out > $O/h2_numerics.txt 2>&1
python tools/scratch/h2_diff.py 2>&1 | grep -v amdgpu > $O/h2_diff.txt
for rep in 1 2; do
  for v in 0 1; do echo "E2E_WG_H2=$v"; E2E_WG_H2=$v python tools/kbench.py L0_64x32 L1_160x64 L2_320x128 L0_32x32d L3_640x256 2>&1 | grep wgrad; done
done > $O/wg_ab.txt 2>&1
cat $O/h2_diff.txt $O/wg_ab.txt; tail -40 $O/h2_numerics.txt
