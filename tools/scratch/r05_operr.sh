cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_operr; mkdir -p $O
for cfg in "" "E2E_CONV_MM=0 E2E_WG_H2=0"; do echo "== $cfg"; env $cfg python tools/scratch/op_err.py 2>&1 | grep -v amdgpu | tail -12; done > $O/op_err.txt 2>&1
cat $O/op_err.txt
