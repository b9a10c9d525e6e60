#!/usr/bin/env python3
"""Long race screen of the LDS-DMA ring kernels (the long form of tests/test_gpu_stress.py).

    python tools/stress_rings.py --reps 200                       # the library in the tree
    MOYOLO_LIB=<another build of libmoyolo.so> MOYOLO_LIB_ALLOW_MISSING=1 python tools/stress_rings.py --reps 200     # A/B on one device
    (round 5's A/B library was the tree at 7b629e6 built with `python -m mo_yolo_amd.build`: profiles/r05_a_*)
    python tools/stress_rings.py --engines 30 --batch 104         # fresh bench-scale engines: NaN-poisoned buffers, repeated passes

Every run of a form is compared BIT FOR BIT with the tiled kernel on the same rows (mo_yolo_amd/stress.py); a bandwidth hog runs on a
second stream and the caches are thrashed between runs.  Prints one JSON line: runs and failures per form.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--forms", default="all")
    ap.add_argument("--dtypes", default="bf16,f16")
    ap.add_argument("--engines", type=int, default=0, help="fresh engines per dtype (poisoned buffers + repeated passes)")
    ap.add_argument("--batch", type=int, default=104)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--no-hog", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--time", type=int, default=0, help="also time every form: mean of this many back-to-back launches on a quiet device (HIP events)")
    a = ap.parse_args()
    import torch
    from mo_yolo_amd import _lib as L
    from mo_yolo_amd import stress as S
    dts = [dict(bf16=torch.bfloat16, f16=torch.float16)[d] for d in a.dtypes.split(",")]
    forms = (S.GEMM_FORMS + S.CONV_FORMS) if a.forms == "all" else ([] if a.forms == "none" else a.forms.split(","))
    res = dict(lib=L.LIB_PATH, device=torch.cuda.get_device_name(0), reps=a.reps, hog=not a.no_hog, forms={}, engines=[])
    t_start = time.time()
    with S.Hog("cuda") as hog:
        for form in forms:
            for dt in dts:
                case = S.gemm_case(form, dt) if form in S.GEMM_FORMS else S.conv_case(form, dt)
                extra = {}
                if form in S.GEMM_FORMS:
                    ref = case.reference()               # tiled kernel, bit-exact reference
                else:
                    ref, close, err = case.base_run()    # the form's own quiet run, held to the tiled kernel within 2 ulp
                    extra = dict(quiet_run_within_2ulp_of_tiled=close, quiet_run_max_abs_vs_tiled=err)
                fails = [] if extra.get("quiet_run_within_2ulp_of_tiled", True) else [dict(run=-1, where="quiet run not within 2 ulp of the tiled kernel")]
                for i in range(a.reps):
                    S.thrash("cuda")
                    if not a.no_hog:
                        hog.kick(4)
                    got = case.run()
                    torch.cuda.synchronize()
                    if not torch.equal(got, ref):
                        fails.append(dict(run=i, where=case.where(got, ref)))
                if a.time:
                    hog.stream.synchronize()
                    torch.cuda.synchronize()
                    out_t = case.run()                  # (allocation of the output outside the timed launches)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.time):
                        case._launch(out_t, 0, case.B if hasattr(case, "B") else case.f["B"])
                    e1.record()
                    torch.cuda.synchronize()
                    extra["us_per_launch"] = round(e0.elapsed_time(e1) / a.time * 1e3, 2)
                    del out_t
                key = f"{form}/{str(dt).replace('torch.', '')}"
                res["forms"][key] = dict(runs=a.reps, failures=len(fails), first=fails[:3], **extra)
                print(f"[stress {time.time() - t_start:6.0f}s] {key}: {len(fails)} of {a.reps} runs differ", file=sys.stderr, flush=True)
                del case, ref
    for dt in dts:
        for e in range(a.engines):
            r = S.engine_determinism(a.config, dt, a.batch, passes=a.passes, hog=not a.no_hog)
            ok = not r["mismatches"] and not r["nan_outputs"] and r["value_planes_vs_tiled"]["equal"]
            r["ok"] = ok
            if not ok or e == 0:
                res["engines"].append(r)
            print(f"[stress {time.time() - t_start:6.0f}s] engine {e} {r['engine']}: {'ok' if ok else 'FAILED ' + json.dumps(r)[:400]}",
                  file=sys.stderr, flush=True)
            torch.cuda.empty_cache()
    res["engines_run"] = a.engines * len(dts)
    res["total_runs"] = sum(v["runs"] for v in res["forms"].values())
    res["total_failures"] = sum(v["failures"] for v in res["forms"].values()) + sum(1 for r in res["engines"] if not r.get("ok", True))
    line = json.dumps(res)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    print(line)
    sys.exit(1 if res["total_failures"] else 0)


if __name__ == "__main__":
    main()
