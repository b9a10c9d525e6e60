#!/usr/bin/env python3
"""Round 6: where a small-batch step spends its time -- kernels or the gaps between them.  Reads a rocprofv3 --kernel-trace csv
(columns Start_Timestamp / End_Timestamp in ns, Kernel_Name), takes the dispatches of the LAST `--steps` replays of a plan of `--launches`
kernels, and prints per step: sum of kernel durations, sum of the gaps between consecutive kernels, wall; plus the launches whose kernel
is longest.   usage: gap_analysis.py <kernel_trace.csv> --launches 80 [--steps 100]"""
import argparse, csv, collections, statistics
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--launches", type=int, required=True); ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.csv))]
rows.sort()
n = a.launches
rows = rows[-n * a.steps:]
ker, gap, wall = [], [], []
per = collections.defaultdict(list)
for s in range(0, len(rows), n):
    st = rows[s:s + n]
    if len(st) < n:
        break
    ker.append(sum(e - b for b, e, _ in st) / 1e3)
    gap.append(sum(max(0, st[i + 1][0] - st[i][1]) for i in range(n - 1)) / 1e3)
    wall.append((st[-1][1] - st[0][0]) / 1e3)
    for i, (b, e, k) in enumerate(st):
        per[(i, k[:70])].append((e - b) / 1e3)
print(f"steps {len(ker)}  launches/step {n}")
print(f"kernel time per step  {statistics.mean(ker):8.1f} us   (min {min(ker):.1f})")
print(f"gaps per step         {statistics.mean(gap):8.1f} us   = {statistics.mean(gap) / (n - 1):.2f} us per gap")
print(f"first start -> last end {statistics.mean(wall):8.1f} us")
top = sorted(per.items(), key=lambda kv: -statistics.mean(kv[1]))[:25]
for (i, k), v in top:
    print(f"  #{i:3d} {statistics.mean(v):7.1f} us  {k}")
short = sum(1 for v in per.values() if statistics.mean(v) < 5.0)
print(f"launches whose kernel runs < 5 us: {short} of {n}")
