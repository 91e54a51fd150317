#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh prof_r3a 2>&1 | tail -16
python bench.py --steps 20 --warmup 3 > gpurun_out/r3_bench_final_a.json 2> gpurun_out/r3_bench_final_a.err
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_final_a.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_secondary']['achieved'], d['forward_only'], d['sliding_window']['seconds'], d['parity']['max_abs_dlogit'], d['parity']['dice_vs_cpu'], d['cpu_baseline']['value'])"
