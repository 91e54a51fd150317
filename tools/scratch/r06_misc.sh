cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_misc; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -s -k "loss_spike" 2>&1 | grep -v "amdgpu.ids\|curr_density" | tail -40 > $O/tests.txt
timeout 1500 python tools/scratch/r06_switch.py 0.15 0.2 0.3 2>&1 | grep -v "amdgpu.ids" > $O/switch2.txt
tail -30 $O/tests.txt; cat $O/switch2.txt
