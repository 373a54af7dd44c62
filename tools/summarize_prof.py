#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/gpu_profiles.sh (<run>/<cfg>/{trace,fetch,write,mfma} + <run>/<cfg>_timeline.txt
+ <run>/bench_*.json) into profiles/<name>_<cfg>_{kernel_stats.csv,pmc.csv,summary.md}.
usage: tools/summarize_prof.py gpurun_out/r2p profiles/round2_v1"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

TITLES = {"c2": "configs[1]: PointNet 1024-pt, 250 instances (`python3 bench.py --no-cpu-baseline --single-mode`)",
          "p32": "configs[2] proxy: one rank's 32-instance shard (`bench.py --instances 32 --no-proxy-full`)",
          "c2f32": "configs[1] with GEOA3_WIDE_MODE=f32 (every convolution on the fp32 MFMA: the bench line's `other_wide_mode`)",
          "c4": "configs[3]: PointNet++ SSG (`bench.py --arch PointNetPP`)",
          "c5": "configs[4]: PointNet 4096-pt, k=32 (`bench.py --npoint 4096 --knn 32`)"}


def counters(path, name):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def one(src, cfg, dst):
    lines = ["# %s" % TITLES.get(cfg, cfg), "",
             "GEOA3_GEO_STREAM=0 (one stream: the default loop overlaps the geometry kernels with the victim's forward on a "
             "second stream, where rocprofv3 interleaves the queues differently and overlapped kernels stretch each other); "
             "the un-profiled bench lines of the default two-stream loop are in `*_bench_*.json`.", ""]
    stats = glob.glob(os.path.join(src, cfg, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(dst + "_kernel_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"]])
        lines += ["## rocprofv3 --kernel-trace --stats", "", "| kernel | calls | avg us | % |", "|---|---|---|---|"]
        for r in rows[:28]:
            lines.append("| `%s` | %s | %.1f | %s |" % (r["Name"].replace("(anonymous namespace)::", "")[:78], r["Calls"],
                                                       float(r["AverageNs"]) / 1e3, r["Percentage"]))
    fe, wr = counters(os.path.join(src, cfg, "fetch"), "FETCH_SIZE"), counters(os.path.join(src, cfg, "write"), "WRITE_SIZE")
    if fe or wr:
        kernels = sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, 0) * 2 + wr.get(k, 0)))
        lines += ["", "## PMC (separate --pmc passes), per launch", "",
                  "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read "
                  "(MI355X_MICROARCH.md, HBM section), so `read MB (x2)` doubles it.", "",
                  "| kernel | FETCH_SIZE KiB | read MB (x2) | WRITE_SIZE KiB | write MB |", "|---|---|---|---|---|"]
        with open(dst + "_pmc.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "FETCH_SIZE_KiB_avg", "read_MB_corrected_x2", "WRITE_SIZE_KiB_avg", "write_MB"])
            for k in kernels[:32]:
                a, b = fe.get(k, 0.0), wr.get(k, 0.0)
                w.writerow([k, "%.1f" % a, "%.1f" % (a * 2 * 1024 / 1e6), "%.1f" % b, "%.1f" % (b * 1024 / 1e6)])
                lines.append("| `%s` | %.0f | %.1f | %.0f | %.1f |" % (k.replace("(anonymous namespace)::", "")[:78], a,
                                                                      a * 2 * 1024 / 1e6, b, b * 1024 / 1e6))
    mf, ga = counters(os.path.join(src, cfg, "mfma"), "SQ_VALU_MFMA_BUSY_CYCLES"), counters(os.path.join(src, cfg, "mfma"), "GRBM_GUI_ACTIVE")
    if mf:
        lines += ["", "## matrix pipe (SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs), per launch", "",
                  "| kernel | MFMA busy SIMD-cycles | GUI active (sum of XCDs) | busy fraction |", "|---|---|---|---|"]
        for k in sorted(mf, key=lambda k: -mf[k])[:8]:
            g = ga.get(k, 0.0)
            lines.append("| `%s` | %.3g | %.3g | %.2f |" % (k.replace("(anonymous namespace)::", "")[:70], mf[k], g,
                                                          mf[k] / 1024.0 / (g / 8.0) if g else 0.0))
    tl = os.path.join(src, cfg + "_timeline.txt")
    if os.path.exists(tl):
        lines += ["", "## one inner iteration (start us, gap to the previous kernel's end, duration)", "", "```"]
        lines += [ln.rstrip()[:150] for ln in open(tl)]
        lines += ["```"]
    open(dst + "_summary.md", "w").write("\n".join(lines) + "\n")
    print("wrote", dst + "_summary.md")


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    for cfg in ("c2", "c2f32", "p32", "c4", "c5"):
        if os.path.isdir(os.path.join(src, cfg)):
            one(src, cfg, "%s_%s" % (dst, cfg))
    for f in glob.glob(os.path.join(src, "bench_*.json")):
        txt = open(f).read().strip()
        if txt.startswith("{"):
            json.loads(txt)
            shutil.copy(f, "%s_%s" % (dst, os.path.basename(f)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
