# round 6 end-of-round fuzz with fresh seeds: K1m shapes (default + small grids), operators, whole networks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz; mkdir -p $O
{
timeout 900 python tools/scratch/fuzz_mm.py 60 611 2>&1 | grep -v "amdgpu.ids" | tail -8
E2E_MM_GRID=8 timeout 900 python tools/scratch/fuzz_mm.py 40 612 2>&1 | grep -v "amdgpu.ids" | tail -8
E2E_MM_GRID=40 timeout 900 python tools/scratch/fuzz_mm.py 40 613 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 900 python tools/scratch/fuzz_pairq.py 60 614 2>&1 | grep -v "amdgpu.ids" | tail -6
timeout 900 python tools/scratch/fuzz_ops.py 80 615 2>&1 | grep -v "amdgpu.ids" | tail -10
timeout 1200 python tools/scratch/fuzz_net.py 16 616 2>&1 | grep -v "amdgpu.ids\|curr_density\|Warning\|cosine" | tail -24
} > $O/out.txt 2>&1
cat $O/out.txt
