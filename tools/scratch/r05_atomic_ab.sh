# K1m data gradient: accumulating destinations by global_atomic_add_f32 (libe2e_hip.so) against load / add / store
# (libe2e_hip_rmw.so = make BUILD=build_rmw LIB=libe2e_hip_rmw.so DEFS=-DMM_ATOMIC_ACC=0): operator + whole-net tests, then the
# training step on one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_atomic; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -k "conv133 or config5 or width48 or whole_net_fixed or config3" 2>&1 | grep -v "curr_density\|amdgpu.ids" | tail -15 > $O/tests.txt
RMW=$PWD/e2enet_medical_amd/csrc/libe2e_hip_rmw.so
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/atomic_$rep.json 2> /dev/null
  E2E_LIB_PATH=$RMW python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/rmw_$rep.json 2> /dev/null
done
python - <<'PY' > gpurun_out/r05_atomic/summary.txt
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_atomic/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s ms/step %.3f  conv family %.3f ms (frac %.4f)  wgrad %.3f ms" % (f.split('/')[-1], d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline']['frac'], d['roofline_secondary']['ms_per_step']))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $O/tests.txt | tail -4; cat $O/summary.txt
