#!/usr/bin/env python3
"""Bit-identity check of the conv_ws kernel variants (MOY_CWS_VARIANT is read once per process, so one process per variant):

    MOY_CONV_WS=2 MOY_CWS_VARIANT=0 python tools/probes/conv_pp_check.py --save /tmp/a.pt
    MOY_CONV_WS=2 MOY_CWS_VARIANT=8 python tools/probes/conv_pp_check.py --save /tmp/b.pt
    python tools/probes/conv_pp_check.py --compare /tmp/a.pt /tmp/b.pt
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

CASES = [(dt, C, B, H, W, res) for dt in ("bf16", "f16") for (C, B, H, W) in ((32, 8, 100, 150), (64, 12, 76, 136), (64, 7, 61, 83), (128, 24, 38, 68), (128, 40, 19, 34), (128, 9, 45, 50))
         for res in (False, True)]


def run():
    from mo_yolo_amd import _lib as L
    from mo_yolo_amd import ops
    outs = {}
    for dtn, C, B, H, W, res in CASES:
        dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtn]
        g = torch.Generator().manual_seed(C * 1000 + H)
        x = (torch.rand(B * H * W, C, generator=g) - 0.5).to(dt)
        w = ((torch.rand(C, 9 * C, generator=g) - 0.5) / math.sqrt(9 * C))
        sc, sh = torch.rand(C, generator=g) * 0.4 + 0.8, (torch.rand(C, generator=g) - 0.5) * 0.2
        rs = (torch.rand(B * H * W, C, generator=g) - 0.5).to(dt)
        buf = torch.zeros(B * H * W, 3 * C + 8, device="cuda", dtype=dt)
        buf[:, :C] = x.cuda()
        buf[:, C:2 * C] = rs.cuda()
        wp = ops.pad_weight(w.cuda(), dt)
        for rep in range(2):            # twice: the second pass runs on a warm ring (catches stale-LDS hazards that depend on timing)
            ops.gemm(buf[:, :C], wp, C, 9 * C, ksize=3, stride=1, geom=(B, H, W, H, W, C), scale=sc.cuda(), shift=sh.cuda(),
                     act=L.ACT_SILU, R=buf[:, C:2 * C] if res else None, out=buf[:, 2 * C:3 * C])
        torch.cuda.synchronize()
        assert float(buf[:, 3 * C:].abs().max()) == 0
        outs[f"{dtn}.C{C}.B{B}.{H}x{W}.res{int(res)}"] = buf[:, 2 * C:3 * C].cpu().clone()
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--compare", nargs=2)
    a = ap.parse_args()
    if a.save:
        torch.save(run(), a.save)
        print("saved", a.save, "variant", os.environ.get("MOY_CWS_VARIANT"))
        return
    x, y = torch.load(a.compare[0]), torch.load(a.compare[1])
    bad = 0
    for k in x:
        same = torch.equal(x[k], y[k])
        d = float((x[k].float() - y[k].float()).abs().max())
        nbad = int((x[k] != y[k]).sum())
        print(f"{k:34s} {'bit-identical' if same else f'DIFFERENT: {nbad} values, max |d| {d:.3e}'}")
        bad += not same
    print("ALL BIT-IDENTICAL" if not bad else f"{bad} case(s) differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
