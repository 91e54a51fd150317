#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; values in KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced streaming
reads (16 B/lane), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv
import re
import glob
import json
import os
import sys
from collections import defaultdict

def summarise(root):
    """{kernel: {launches, fetch_bytes_per_launch_corrected, write_bytes_per_launch, hbm_bytes_per_launch}, "_meta": ...} from
    <root>/pmc_fetch and <root>/pmc_write (rocprofv3 --pmc output directories), sorted by total bytes."""
    tot = defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_fetch": 0, "n_write": 0})
    for sub, key, cnt in (("pmc_fetch", "FETCH_SIZE", "n_fetch"), ("pmc_write", "WRITE_SIZE", "n_write")):
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] != key:
                        continue
                    m = re.search(r"(?:::)?([A-Za-z_][A-Za-z0-9_]*(?:<[^>]*>)?)\((?!anonymous)", row["Kernel_Name"])
                    name = m.group(1) if m else row["Kernel_Name"][:70]
                    tot[name][key] += float(row["Counter_Value"])
                    tot[name][cnt] += 1
    out = {}
    for name, v in tot.items():
        if not v["n_fetch"]:
            continue
        fetch = 2.0 * v["FETCH_SIZE"] * 1024 / v["n_fetch"]
        write = v["WRITE_SIZE"] * 1024 / max(v["n_write"], 1)
        out[name] = {"launches": v["n_fetch"], "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
                     "hbm_bytes_per_launch": fetch + write}
    rows = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    doc = dict(rows)
    # which library the counters were collected from: bench.py refuses a summary whose ABI version is not the loaded library's
    # (kernels changed since: the numbers would be stale)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from e2enet_medical_amd._lib import ABI_VERSION
        doc["_meta"] = {"abi_version": ABI_VERSION}
    except Exception:
        pass
    return doc


if __name__ == "__main__":
    root = sys.argv[1]
    doc = summarise(root)
    json.dump(doc, open(os.path.join(root, "traffic.json"), "w"), indent=1)
    for name, v in [kv for kv in doc.items() if kv[0] != "_meta"][:14]:
        print("%-72s n=%-4d fetch=%8.1f MB write=%8.1f MB" % (name, v["launches"], v["fetch_bytes_per_launch_corrected"] / 1e6,
                                                              v["write_bytes_per_launch"] / 1e6))
