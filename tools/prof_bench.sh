#!/bin/bash
# usage (GPU box, repo root): tools/prof_bench.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of bench.py as it runs by default (deep levels and weight gradients on extra HIP streams)
# 2) the same with everything issued on one stream (E2E_LANES=0 E2E_WGRAD_STREAM=0): per-kernel durations without co-runners,
#    the ones bench.py's instrumented steps measure with HIP events
# 3) two separate --pmc passes (FETCH_SIZE / WRITE_SIZE), one stream
TAG=$1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/$TAG/bench_trace.json 2> $R/gpurun_out/$TAG/trace.err
export E2E_LANES=0 E2E_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace_serial -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/$TAG/bench_trace_serial.json 2> $R/gpurun_out/$TAG/trace_serial.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/$TAG/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/$TAG/pmc_write.err
cd $R
python3 tools/traffic_summary.py gpurun_out/$TAG
for t in trace trace_serial; do
  f=$(find gpurun_out/$TAG/$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/$TAG/${t}_kernel_stats.csv
done
