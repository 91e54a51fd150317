#!/bin/bash
# The one GPU-box runner (replaces the per-experiment r3_*.sh scripts).  From the repo root on the GPU box:
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [tests] [bench] [prof] [kbench "<kbench args>"]'
#
#   tests   python -m pytest tests -m gpu -x -q                          -> gpurun_out/<tag>/tests.log
#   ktests "<expr>"   the same with -k <expr>                            -> gpurun_out/<tag>/ktests.log
#   bench   python bench.py --steps 20 --warmup 5                        -> gpurun_out/<tag>/bench.json
#   benchlive  the same with --live-traffic (roofline.traffic collected by the run itself)  -> gpurun_out/<tag>/bench_live.json
#   prof    tools/prof_bench.sh <tag> (rocprofv3 --kernel-trace --stats, default + one-stream; FETCH_SIZE / WRITE_SIZE passes)
#   kbench  python tools/kbench.py <args> (single-kernel timings)        -> gpurun_out/<tag>/kbench.log
#
# Counter passes for single kernels: tools/pmc.sh, tools/pmc_lds.sh, tools/pmc_mfma.sh, tools/pmc_one.sh (each documents itself).
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ (tracked) afterwards.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
while [ $# -gt 0 ]; do
  case $1 in
    tests)  timeout 1500 python -m pytest tests -m gpu -q --tb=short -rs 2>&1 | grep -v "curr_density\|amdgpu.ids" | tail -150 > gpurun_out/$TAG/tests.log ;;
    ktests) shift; timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -k "$1" 2>&1 | grep -v "curr_density\|amdgpu.ids" | tail -150 > gpurun_out/$TAG/ktests.log ;;
    bench)  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ;;
    benchlive) timeout 1500 python bench.py --steps 20 --warmup 5 --live-traffic > gpurun_out/$TAG/bench_live.json 2> gpurun_out/$TAG/bench_live.err ;;
    prof)   bash tools/prof_bench.sh $TAG 2>&1 | tail -20 > gpurun_out/$TAG/prof.log ;;
    kbench) shift; timeout 600 python tools/kbench.py $1 > gpurun_out/$TAG/kbench.log 2>&1 ;;
  esac
  shift
done
tail -5 gpurun_out/$TAG/*.log 2>/dev/null
