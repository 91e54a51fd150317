# round-5 evidence on the final code: parity record, rocprof stats + traffic, K1m counters, one-rank RCCL self-test of the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z
timeout 1500 python tools/parity_report.py --out gpurun_out/r05z/parity.json --tag final --net128 > gpurun_out/r05z/parity.txt 2>&1
bash tools/prof_bench.sh r05z > gpurun_out/r05z/prof.log 2>&1
bash tools/scratch/r05_mm_pmc.sh > gpurun_out/r05z/mm_pmc.log 2>&1
E2E_FORCE_DIST=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r05z/bench_force_dist.json 2> gpurun_out/r05z/bench_force_dist.err
tail -25 gpurun_out/r05z/parity.txt; tail -3 gpurun_out/r05z/prof.log; head -c 1500 gpurun_out/r05z/bench_force_dist.json
