# K1m epilogue: 16-byte stores after a quad transpose (libe2e_hip.so) against 4-byte stores (libe2e_hip_dword.so = commit 91219f8), one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_store; mkdir -p $O
A=$PWD/e2enet_medical_amd/csrc/libe2e_hip_dword.so
{
for rep in 1 2 3; do
  echo "== dword stores (91219f8), pass $rep"; E2E_LIB_PATH=$A KB_ITERS=40 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "fwd\|dgrad"
  echo "== transposed 16-byte stores, pass $rep"; KB_ITERS=40 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 2>&1 | grep "fwd\|dgrad"
done
} > $O/kbench.txt 2>&1
for rep in 1 2; do
  E2E_LIB_PATH=$A python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/dword_$rep.json 2> /dev/null
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/x4_$rep.json 2> /dev/null
done
python - <<'PY' > gpurun_out/r05_store/summary.txt
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_store/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-20s ms/step %.3f  conv family %.3f ms (frac %.4f)  wgrad %.3f ms  clock %.0f" % (f.split('/')[-1], d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline']['frac'], d['roofline_secondary']['ms_per_step'], d['roofline']['measured_clock_mhz']))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $O/kbench.txt; cat $O/summary.txt
