# round 5: first run of conv133_mm_kernel: operator tests, then timings against the round-4 kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_mm; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --tb=short -k "conv133_fwd_bwd or h2_and_bf3 or split_operand" 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/tests.log
for rep in 1 2; do
  python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 L1_64x64d 2>&1 | grep -v amdgpu
  KB_NO_MM=1 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 L1_64x64d 2>&1 | grep -v amdgpu
done > $O/kbench.txt 2>&1
tail -30 $O/tests.log; cat $O/kbench.txt
