#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) per kernel.

usage: pmc_summary.py <dir with one sub-dir per pass> <out.json>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (MI355X_MICROARCH.md, HBM section); the
span encoder's loads are narrow scattered reads, for which the guide's 2x "wide coalesced stream"
correction does not apply -- values are reported uncorrected and flagged as such.
"""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
res = {}
for k, d in agg.items():
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0].strip()[:80] if not short.startswith("rocprim") else "rocprim::radix_sort_onesweep(*)"
    e = {c: v for c, v in d.items()}
    e["dispatches"] = max(len(disp[(k, c)]) for c in d)
    res.setdefault(short, {}).update(e)
for k, e in res.items():
    n = e["dispatches"]
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_dispatch_uncorrected"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0 / n
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    print(k, {c: (int(v) if isinstance(v, float) else v) for c, v in e.items()})
