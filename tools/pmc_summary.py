#!/usr/bin/env python
"""Summarise rocprofv3 --pmc csv passes: per kernel name, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")[:90]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in sorted(acc.items(), key=lambda kv: -len(kv[1])):
    if "conv" not in name and "Conv" not in name:
        continue
    print(name)
    for c, v in sorted(ctrs.items()):
        print("   %-26s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
