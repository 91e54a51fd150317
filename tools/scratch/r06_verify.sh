cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_verify; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_ops.py -q -m gpu -k "loss_spike or conv133_fwd_bwd or config5 or alternative_paths" 2>&1 | grep -v "amdgpu.ids\|curr_density" | tail -8 > $O/tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "amdgpu.ids\|curr_density" | tail -4 > $O/smoke.txt
cat $O/tests.txt $O/smoke.txt
