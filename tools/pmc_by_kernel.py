#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: tools/pmc_by_kernel.py <dir> [kernel substring]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in agg.items():
    if pat in k:
        print(k[:90], {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", max(len(v) for v in cs.values()))
