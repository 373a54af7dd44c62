#!/usr/bin/env python3
"""Timeline of ONE inner iteration from a rocprofv3 --kernel-trace CSV (start offset, gap to the previous kernel's end,
duration, grid, registers, LDS): tools/trace_timeline.py <dir-or-csv> [iteration index from the end, default 3]"""
import csv
import glob
import os
import sys


def main(src, back=3):
    f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "attack_update" in r["Kernel_Name"]]
    a, b = idx[-back - 1], idx[-back]
    t0 = prev = int(rows[a]["End_Timestamp"])
    tot = gaps = 0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print("%8.1f gap %6.1f dur %7.1f  %-52s grid %sx%sx%s wg %s vgpr %s+%s lds %s scratch %s q%s" % (
            (s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:52], r["Grid_Size_X"], r["Grid_Size_Y"],
            r["Grid_Size_Z"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"],
            r["Scratch_Size"], r["Queue_Id"]))
        tot += e - s
        gaps += max(0, s - prev)
        prev = max(prev, e)
    print("kernels %.1f us, gaps %.1f us, span %.1f us, launches %d" % (tot / 1e3, gaps / 1e3, (prev - t0) / 1e3, b - a))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
