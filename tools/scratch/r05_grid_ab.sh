# K1m grid below the CU count: CUs left to the kernels of the other streams while a persistent K1m launch holds the rest
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_grid; mkdir -p $O; rm -f $O/*.txt
for rep in 1 2; do
for g in 256 248 240 224; do
  E2E_MM_GRID=$g python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E2E_MM_GRID=$g', 'ms/step %.3f' % d['ms_per_step'], 'conv family %.3f' % d['roofline']['ms_per_step'], 'clocks', round(d['roofline']['measured_clock_mhz']), round(d['roofline_secondary']['measured_clock_mhz']))" >> $O/out.txt
done
done
cat $O/out.txt
