#!/usr/bin/env python3
"""Print per-kernel average durations from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
pat = sys.argv[2:] or [""]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in pat):
        print("%-80s calls=%5s avg=%9.1f us %6s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
