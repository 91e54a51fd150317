# config 5 (AMOS-shaped, K = 16) gradients under the engine's own branch decisions, per kernel family switched off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_cfg5f; mkdir -p $O
for envs in "X=1" "E2E_CONV_MM=0" "E2E_WG_H2=0" "E2E_CONV_MM=0 E2E_WG_H2=0" "E2E_CONV_MM=0 E2E_WG_H2=0 E2E_WG_BF3=0 E2E_CT_BF3=0 E2E_CONV_DENSE=0"; do
  echo "== $envs"
  env $envs timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q --tb=line -k "config5 and 0.1" -s 2>&1 | grep "grad, same\|grad noise\|passed\|failed"
done > $O/out.txt 2>&1
cat $O/out.txt
