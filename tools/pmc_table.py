#!/usr/bin/env python3
"""One line per kernel from a tools/pmc_plan.sh / tools/pmc_gemm.sh run: matrix pipe busy, vector : matrix instructions, LDS activity, waits.
   busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)   (the convention of DESIGN.md round 2: 55.9 % for the C = 128 3x3 kernel)
usage: pmc_table.py <dir>"""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        if "moy::" not in k:
            continue
        agg[(k, row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for (k, grid), d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = max(len(v) for v in d.values())
    busy, wc = m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAVE_CYCLES", 0)
    name = k.replace("void moy::", "").split("(moy")[0].split("(float")[0][:64]
    rows.append((wc, name, grid, n,
                 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32 * busy) if busy else 0,
                 m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"] if m.get("SQ_INSTS_MFMA") else float("nan"),
                 m.get("SQ_INSTS_LDS", 0) / m["SQ_INSTS_MFMA"] if m.get("SQ_INSTS_MFMA") else float("nan"),
                 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else 0,
                 100 * m.get("SQ_WAIT_ANY", 0) / wc if wc else 0, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0,
                 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc if wc else 0, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else 0))
rows.sort(key=lambda r: -r[0] * r[3])
print(f"{'kernel':64s} {'grid':>9s} {'n':>3s} {'mfma busy%':>10s} {'valu/mfma':>9s} {'lds/mfma':>8s} {'bank cfl%':>9s} {'wait any%':>9s} {'wait inst%':>10s} {'wait lds%':>9s} {'valu act%':>9s}")
for r in rows:
    print(f"{r[1]:64s} {r[2]:>9s} {r[3]:3d} {r[4]:10.1f} {r[5]:9.2f} {r[6]:8.2f} {r[7]:9.1f} {r[8]:9.1f} {r[9]:10.1f} {r[10]:9.1f} {r[11]:9.1f}")
