#!/usr/bin/env python3
"""What bounds the value projection (gemm_wreg_kernel, M = 288 x 13566, N = 1536, K = 256, head planes)?  Timing-only builds of the
SAME kernel with one of its three activities removed (MOY_WREG_ABL: bit 0 no MFMAs, bit 1 output stores dropped by the descriptor's
range check, bit 2 no activation DMA past the prologue, bit 4 s_memtime stamps per phase), one child process per build (the switch is read once), plus a plain fill
and copy of the same 12 GB / 2 GB on this device.  Results are garbage in every build but 0.

usage: wreg_ablate.py [frames=288]"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child(frames):
    import torch
    from mo_yolo_amd import ops
    dev, dt = "cuda", torch.bfloat16
    torch.manual_seed(5)
    M, N, K = frames * 13566, 1536, 256
    x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
    w = ops.pad_weight((torch.rand(N, K, device=dev) - 0.5) / 16, dt)
    sh = torch.rand(N, device=dev) - 0.5
    out = torch.empty(N // 32, M, 32, device=dev, dtype=dt)
    f = lambda: ops.gemm(x, w, N, K, out=out[0], shift=sh, planes=(32, M * 32))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(4):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    res = dict(ms=min(ts), ms_all=[round(t, 4) for t in ts])
    v = out.view(torch.int16).view(-1)
    res["checksum"] = [int(v[i::7].to(torch.int64).sum().item()) for i in range(2)] + [int((v.to(torch.int32) * 31 % 1009).sum().item())]
    if int(os.environ.get("MOY_WREG_ABL", "0")) >= 16:
        st = out.view(torch.int64).view(-1)[:16].tolist()
        n = max(1, st[5])
        res["stamps_cycles_per_tile(issue,mfma,epilogue,vmwait,barrier)"] = dict(wave0=[round(x / n) for x in st[:5]], wave4=[round(x / n) for x in st[8:13]], tiles=st[5])
    if os.environ.get("MOY_WREG_ABL", "0") == "0":
        flat = out.view(-1)
        for name, g in (("fill_12GB", lambda: flat.fill_(1.0)), ("copy_2GB_to_2GB", lambda: flat[:M * K].copy_(x.view(-1))),
                        ("read_2GB_sum", lambda: x.view(torch.int16).view(-1)[: M * K].sum())):
            g()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                g()
            e1.record()
            torch.cuda.synchronize()
            res[name + "_ms"] = e0.elapsed_time(e1) / 3
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if os.environ.get("WREG_ABLATE_CHILD"):
        child(int(sys.argv[1]))
        sys.exit(0)
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 288
    names = {0: "full", 1: "no MFMA", 2: "no stores", 4: "no DMA", 3: "DMA only", 5: "stores only", 6: "MFMA only",
             16: "stamps", 22: "stamps MFMA only"}
    which = [int(x) for x in os.environ.get("WREG_ABLATE_SET", "0,1,2,4,3,5,6").split(",")]
    for rnd in range(2):
        for abl in which:
            env = dict(os.environ, WREG_ABLATE_CHILD="1", MOY_WREG_ABL=str(abl))
            r = subprocess.run([sys.executable, __file__, str(frames)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            print(f"round {rnd} abl {abl} ({names[abl]:11s}):", line[0][7:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
