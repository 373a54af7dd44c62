#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<run>/{trace,fetch,write}) into profiles/<name>_*.csv|md.
usage: tools/summarize_prof.py gpurun_out/r1 profiles/round1"""
import collections
import csv
import glob
import os
import sys


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    lines = []
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(dst + "_kernel_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"]])
        lines.append("## rocprofv3 --kernel-trace --stats (python3 bench.py)\n")
        lines.append("| kernel | calls | avg us | % |\n|---|---|---|---|")
        for r in rows[:24]:
            lines.append("| `%s` | %s | %.1f | %s |" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       r["Percentage"]))
    pmc = {}
    for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pmc[name] = {k: sum(v) / len(v) for k, v in agg.items()}
    if pmc:
        kernels = sorted(set().union(*[set(v) for v in pmc.values()]),
                         key=lambda k: -(pmc.get("FETCH_SIZE", {}).get(k, 0) + pmc.get("WRITE_SIZE", {}).get(k, 0)))
        lines.append("\n## PMC (separate --pmc passes), per launch\n")
        lines.append("FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read "
                     "(MI355X_MICROARCH.md, HBM section), so `read MB (x2)` doubles it.\n")
        lines.append("| kernel | FETCH_SIZE KiB | read MB (x2) | WRITE_SIZE KiB | write MB |\n|---|---|---|---|---|")
        with open(dst + "_pmc.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "FETCH_SIZE_KiB_avg", "read_MB_corrected_x2", "WRITE_SIZE_KiB_avg", "write_MB"])
            for k in kernels[:30]:
                fe, wr = pmc.get("FETCH_SIZE", {}).get(k, 0.0), pmc.get("WRITE_SIZE", {}).get(k, 0.0)
                w.writerow([k, "%.1f" % fe, "%.1f" % (fe * 2 * 1024 / 1e6), "%.1f" % wr, "%.1f" % (wr * 1024 / 1e6)])
                lines.append("| `%s` | %.0f | %.1f | %.0f | %.1f |" % (k[:70], fe, fe * 2 * 1024 / 1e6, wr,
                                                                      wr * 1024 / 1e6))
    p = os.path.join(src, "trace.log")
    if os.path.exists(p):
        for ln in open(p):
            if ln.startswith('{"metric'):
                lines.append("\n## bench line of the profiled run\n\n```json\n%s```" % ln)
    open(dst + "_summary.md", "w").write("\n".join(lines) + "\n")
    print("wrote", dst + "_summary.md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
