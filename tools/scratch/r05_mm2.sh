# round 5: conv133_mm_kernel: operator tests, timings, phase stamps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_mm; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --tb=short -k "conv133_fwd_bwd" 2>&1 | grep -v "amdgpu.ids" | tail -15 > $O/tests.log
for rep in 1 2; do python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 L1_64x64d 2>&1 | grep -v "amdgpu\|wgrad"; done > $O/kbench.txt 2>&1
bash tools/scratch/r05_mm_stamps.sh > $O/stamps_out.txt 2>&1
tail -4 $O/tests.log; cat $O/kbench.txt; grep "conv133_mm mode" $O/stamps_out.txt
