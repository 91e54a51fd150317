#!/bin/bash
# round 3, first GPU call: measured parity (default and two-level builds) + per-layer step profile of both
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/parity_report.py --out gpurun_out/r03_parity.json --tag default --net128 > gpurun_out/r3_parity_default.log 2>&1
E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_twolvl.so python tools/parity_report.py --out gpurun_out/r03_parity.json --tag twolvl_all > gpurun_out/r3_parity_twolvl.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --op-profile > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_twolvl.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --op-profile > gpurun_out/r3_bench_twolvl.json 2> gpurun_out/r3_bench_twolvl.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_bench_default2.json 2>/dev/null
tail -30 gpurun_out/r3_parity_default.log
tail -20 gpurun_out/r3_parity_twolvl.log
