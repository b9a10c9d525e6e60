#!/usr/bin/env python3
"""Per-launch-shape kernel durations from a rocprofv3 --kernel-trace csv: groups by (kernel name, grid size), so that one
template instantiation serving several shapes (e.g. gemm_wreg_kernel: value projection and 1x1 convs) is reported per shape.
usage: trace_by_grid.py <dir with *_kernel_trace.csv> [min_calls]"""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[(r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(d)
tot = sum(sum(v) for v in agg.values())
print("kernel,grid,calls,total_us,avg_us,min_us,max_us,percent")
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"\"{k[:110]}\",{g},{len(v)},{sum(v)/1e3:.1f},{sum(v)/len(v)/1e3:.1f},{min(v)/1e3:.1f},{max(v)/1e3:.1f},{100*sum(v)/tot:.2f}")
