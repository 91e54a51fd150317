# the stream layout on the final kernels: weight-gradient stream and lanes on / off, one box, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_streams; mkdir -p $O; rm -f $O/*.txt
for rep in 1 2; do
for cfg in "E2E_WGRAD_STREAM=1 E2E_LANES=1" "E2E_WGRAD_STREAM=0 E2E_LANES=1" "E2E_WGRAD_STREAM=1 E2E_LANES=0" "E2E_WGRAD_STREAM=0 E2E_LANES=0"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms/step %.3f' % d['ms_per_step'], 'clocks', round(d['roofline']['measured_clock_mhz']), round(d['roofline_secondary']['measured_clock_mhz']))" >> $O/out.txt
done
done
cat $O/out.txt
