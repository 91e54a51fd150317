#!/bin/bash
cd $GRAFT_REPO_ROOT
E2E_BENCH_ALL_LAUNCHES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --op-profile > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench8.json'))
from collections import defaultdict
agg=defaultdict(lambda:[0,0.0])
for ms,k,ints in d["all_launches"]:
    if k.startswith("convT") or k.startswith("maxpool") or k.startswith("head") or k.startswith("in_"):
        key=(k,tuple(ints[:9]))
        agg[key][0]+=1; agg[key][1]+=ms
for (k,ints),(n,ms) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:60]:
    print("%-14s %-44s n=%d total %.3f ms (%.3f each)"%(k,ints,n,ms,ms/n))
PY
