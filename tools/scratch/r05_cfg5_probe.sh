cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_cfg5p; mkdir -p $O
{ echo "== config5 d=0.1 test with one stream (E2E_LANES=0 E2E_WGRAD_STREAM=0 E2E_GRAPHS=0)"
E2E_LANES=0 E2E_WGRAD_STREAM=0 E2E_GRAPHS=0 timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q --tb=line -k "config5 and 0.1" -s 2>&1 | grep "grad, same\|passed\|failed"
echo "== per-op probe"
timeout 600 python tools/scratch/mm_dgrad_probe.py 0.1 2>&1 | grep -v "curr_density\|amdgpu.ids\|Total"
} > $O/out.txt 2>&1
cat $O/out.txt | cut -c1-400
