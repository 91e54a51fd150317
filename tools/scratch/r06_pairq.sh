cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pairq; mkdir -p $O
{
timeout 600 python tools/scratch/fuzz_pairq.py 40 1 2>&1 | grep -v "amdgpu.ids"
E2E_MM_GRID=8 timeout 600 python tools/scratch/fuzz_pairq.py 40 2 2>&1 | grep -v "amdgpu.ids"
E2E_MM_GRID=24 timeout 600 python tools/scratch/fuzz_pairq.py 40 3 2>&1 | grep -v "amdgpu.ids"
echo "--- kbench, E2E_MM_PAIRQ=1 (default) then 0, interleaved twice"
for r in 1 2; do
  timeout 300 python tools/kbench.py L0_64x32 L0_96x32 L0_128x32 L0_160x32 2>&1 | grep -v "amdgpu.ids" | grep "\[mm\]"
  E2E_MM_PAIRQ=0 timeout 300 python tools/kbench.py L0_64x32 L0_96x32 L0_128x32 L0_160x32 2>&1 | grep -v "amdgpu.ids\|unknown E2E" | grep "\[mm\]" | sed 's/^/PAIRQ=0 /'
done
} > $O/out.txt 2>&1
tail -40 $O/out.txt
timeout 600 python -m pytest tests/test_gpu_net.py -q -m gpu -k "packed_weights" 2>&1 | tail -5 >> $O/out.txt
tail -8 $O/out.txt
