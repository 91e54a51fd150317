cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fuzz; mkdir -p $O
{
timeout 900 python tools/scratch/fuzz_mm.py 60 1 2>&1 | grep -v "amdgpu.ids"
E2E_MM_GRID=8 timeout 900 python tools/scratch/fuzz_mm.py 60 2 2>&1 | grep -v "amdgpu.ids"
E2E_MM_GRID=24 timeout 900 python tools/scratch/fuzz_mm.py 60 3 2>&1 | grep -v "amdgpu.ids"
timeout 600 python tools/scratch/fuzz_ops.py 60 7 2>&1 | grep -v "amdgpu.ids"
} > $O/out.txt 2>&1
tail -30 $O/out.txt
