#!/usr/bin/env python3
"""HBM traffic from the two PMC passes of tools/profile_round.sh (or tools/pmc_traffic.sh): per kernel and for the whole plan.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> doubled;
WRITE_SIZE is exact for 16-B-per-lane stores.  Units: KB per dispatch.   hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024.

usage: traffic_summary.py <dir with fetch/ and write/> [--steps N --frames F]
With --steps/--frames: the sum over the library's kernels (namespace moy) of all N steps / (N * F) = measured HBM bytes per frame,
written to <dir>/traffic_step.json (copied into profiles/traffic_by_launch.json: "step_total")."""
import argparse, collections, csv, glob, json
ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--steps", type=int, default=0)
ap.add_argument("--frames", type=int, default=0)
a = ap.parse_args()
root = a.root


def load(sub, counter):
    d = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                d[(r["Kernel_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    return d


fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
rows = []
for k in fe:
    f = sum(fe[k]) / len(fe[k])
    w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
    rows.append(dict(kernel=k[0][:140], grid=k[1], dispatches=len(fe[k]), fetch_kb_raw=f, write_kb=w, hbm_bytes=(2 * f + w) * 1024,
                     ours=("moy::" in k[0])))
rows.sort(key=lambda r: -r["hbm_bytes"] * r["dispatches"])
json.dump(rows, open(root + "/traffic.json", "w"), indent=1)
for r in rows[:14]:
    print(f"{r['hbm_bytes']/1e6:10.1f} MB  grid {r['grid']:>9s} x{r['dispatches']:3d}  {r['kernel'][:90]}")
if a.steps and a.frames:
    # launches a plan makes once when it is built (the row-run score plan: the masked-token pass and its self-check) are not part
    # of a step: per launch shape, only the last (count - count % steps) dispatches are the steps'.
    def step_part(d):
        return {k: v[len(v) % a.steps:] for k, v in d.items() if "moy::" in k[0]}
    sf, sw = step_part(fe), step_part(wr)
    setup = sum(len(v) % a.steps for k, v in fe.items() if "moy::" in k[0])
    tot_f = sum(sum(v) for v in sf.values())
    tot_w = sum(sum(v) for v in sw.values())
    other = sum(sum(v) for k, v in fe.items() if "moy::" not in k[0]) * 2 + sum(sum(v) for k, v in wr.items() if "moy::" not in k[0])
    n_disp = sum(len(v) for v in sf.values())
    assert n_disp % a.steps == 0, f"{n_disp} dispatches are not {a.steps} passes of one plan"
    per_frame = (2 * tot_f + tot_w) * 1024 / (a.steps * a.frames)
    doc = dict(hbm_bytes_per_frame=per_frame, fetch_kb_raw_total=tot_f, write_kb_total=tot_w, steps=a.steps, frames_per_step=a.frames,
               dispatches=n_disp, dispatches_per_step=n_disp / a.steps, setup_dispatches_excluded=setup, other_kernels_bytes=other * 1024,
               formula="(2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the library's dispatches / (steps*frames)")
    json.dump(doc, open(root + "/traffic_step.json", "w"), indent=1)
    print(json.dumps(doc))
