#!/usr/bin/env python3
"""Copy the judged summaries of one tools/profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
usage: collect_profiles.py gpurun_out/<dir> <tag>      e.g.  collect_profiles.py gpurun_out/r02p r02_a
Also refreshes profiles/traffic_by_launch.json (read by bench.py for roofline.traffic / roofline_step.traffic_frac)."""
import glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")


def cp(a, b):
    if os.path.exists(a):
        shutil.copyfile(a, os.path.join(dst, b))
        print("  ", b)


for f in sorted(glob.glob(f"{src}/bench_*.json")):
    cp(f, f"{tag}_{os.path.basename(f)}")
for f in sorted(glob.glob(f"{src}/launches_*.json")):
    cp(f, f"{tag}_launch_table_{os.path.basename(f)[9:]}")
for sub, name in (("trace_default", "default_2streams"), ("trace_1stream", "b288_1stream")):
    for f in glob.glob(f"{src}/{sub}/**/*kernel_stats.csv", recursive=True):
        cp(f, f"{tag}_kernel_stats_{name}.csv")
cp(f"{src}/trace_1stream_by_launch_shape.csv", f"{tag}_b288_1stream_by_launch_shape.csv")
cp(f"{src}/traffic.json", f"{tag}_b288_hbm_traffic_pmc.json")
cp(f"{src}/traffic_step.json", f"{tag}_b288_hbm_traffic_step.json")

tp = os.path.join(dst, "traffic_by_launch.json")
doc = json.load(open(tp)) if os.path.exists(tp) else {}
if "launches" not in doc:
    doc = {"launches": doc, "step_total": {}}
if os.path.exists(f"{src}/traffic_step.json") and os.path.exists(f"{src}/traffic.json"):
    st = json.load(open(f"{src}/traffic_step.json"))
    doc["step_total"]["c2_bf16"] = dict(hbm_bytes_per_frame=st["hbm_bytes_per_frame"], dispatches_per_pass=st["dispatches_per_step"],
                                        source=f"profiles/{tag}_b288_hbm_traffic_step.json: {st['formula']}; {st['steps']} passes of the "
                                               f"plan at {st['frames_per_step']} frames, one engine, one stream, eager launches")
    rows = json.load(open(f"{src}/traffic.json"))
    for r in rows:
        # the dominant launch.  Round 4 (input_proj folded): the value projection of the P3 level, K = 128, 64-row tiles -- the only
        # dispatch of that template per pass.  Rounds 2-3: one value projection over all tokens, K = 256 (kept for old trace dirs).
        key = None
        if "gemm_wreg_kernel" in r["kernel"] and "64, 3, 1, false, 8, 64, 128" in r["kernel"] and r["grid"] == "122880":
            key = "gemm1x1 M2976768 N1536 K128"
        elif "gemm_wreg_kernel" in r["kernel"] and "32, 3, 1, false, 8, 64, 256" in r["kernel"] and r["grid"] == "122880" and r["dispatches"] == 5:
            key = "gemm1x1 M3907008 N1536 K256"
        if "stem_l1_kernel" in r["kernel"] and r["dispatches"] == 5:      # within 3 % of the value launch in time: whichever is longer on a device is `roofline`
            key = "stem+conv1 fused M11907072 N64"
        elif "c2f_fused_kernel" in r["kernel"] and r["dispatches"] == 5:
            key = "c2f fused M11907072 64->[32|32]->64"
        elif ("msda_raw_kernel" in r["kernel"] or "msda_raw_mfma_kernel" in r["kernel"]) and r["dispatches"] == 30:      # round 5: six calls per pass
            key = "msda_raw0 M86400"
        elif "conv_ws_kernel" in r["kernel"] and "128, 128, 8, 8, 2, false" in r["kernel"] and r["dispatches"] == 30:
            # the 3x3 stride-1 convolutions with 128 channels at the P4 level (six plain calls per pass; the two with the shortcut are a template
            # and a launch name of their own): the runner-up of the raw gather by total time (round 5)
            key = "gemm3x3 M744192 N128 K1152"
        if key:
            doc["launches"][key] = dict(
                hbm_bytes=r["hbm_bytes"], source=f"profiles/{tag}_b288_hbm_traffic_pmc.json ({r['kernel'][:70]}, grid {r['grid']}, "
                f"{r['dispatches']} dispatches): (2*FETCH_SIZE + WRITE_SIZE)*1024, fetch {r['fetch_kb_raw']:.0f} KB raw, write {r['write_kb']:.0f} KB")
    json.dump(doc, open(tp, "w"), indent=1)
    print("   traffic_by_launch.json: c2_bf16", round(st["hbm_bytes_per_frame"] / 1e6, 1), "MB/frame")
