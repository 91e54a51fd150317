cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_graderr; mkdir -p $O
python tools/scratch/h2_nan.py 2>&1 | grep -v amdgpu > $O/h2_nan.txt
for cfg in "" "E2E_CONV_MM=0" "E2E_WG_H2=0" "E2E_CONV_MM=0 E2E_WG_H2=0"; do
  echo "== $cfg"; env $cfg python tools/scratch/grad_err.py amos 2>&1 | grep -v amdgpu | tail -14
done > $O/graderr.txt 2>&1
cat $O/h2_nan.txt; cat $O/graderr.txt
