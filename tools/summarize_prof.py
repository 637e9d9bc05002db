#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time stats and per-dispatch counters."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.split("(")[0][:70]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (short(row.get("Name", "")), row.get("Calls"),
              row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))

for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    d = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        n = short(row["Kernel_Name"])
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[n] = {k: row.get(k) for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                           "Workgroup_Size", "Grid_Size", "Accum_VGPR_Count")}
    print("== kernel trace:", os.path.relpath(f, out))
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print("  %-70s n=%d avg=%.1fus med=%.1fus min=%.1fus max=%.1fus %s" % (
            n, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3, v2[-1] / 1e3, meta[n]))

for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters:", os.path.relpath(f, out))
    for n, cs in acc.items():
        if "align_fused" not in n and "accumulate" not in n:
            continue
        for c, v in cs.items():
            print("  %-50s %-24s n=%d avg=%.6g" % (n[:50], c, len(v), sum(v) / len(v)))
