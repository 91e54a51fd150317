cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_cfg5; mkdir -p $O
for cfg in "" "E2E_CONV_MM=0" "E2E_WG_H2=0"; do
  echo "== $cfg"; env $cfg python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "config5 or width48_whole" 2>&1 | grep "grad noise\|logit parity\|passed\|failed\|Error\|width 48"
done > $O/cfg5.txt 2>&1
cat $O/cfg5.txt
