#!/usr/bin/env python3
"""Bit-identity check of two builds of moy_decoder_tail (MOYOLO_LIB selects the library: one process per build):
    MOYOLO_LIB=.../libmoyolo_base.so python tools/probes/tail_check.py --save /tmp/a.pt
    python tools/probes/tail_check.py --save /tmp/b.pt ;  python tools/probes/tail_check.py --compare /tmp/a.pt /tmp/b.pt"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def run():
    from mo_yolo_amd import ops
    outs = {}
    for dtn in ("bf16", "f16"):
        dt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtn]
        for M, dffn in ((300 * 7 + 13, 1024), (128, 256), (86400, 1024)):
            g = torch.Generator().manual_seed(M + dffn)
            r = lambda *s, sc=1.0: ((torch.rand(*s, generator=g) - 0.5) * sc).cuda()
            pw = lambda w: ops.pad_weight(w, dt)
            samp, e1 = r(M, 256).to(dt), r(M, 256).to(dt)
            vec = lambda n=256: r(n, sc=0.2)
            ref = torch.rand(M, 4, generator=g).cuda()
            args = (samp, e1, pw(r(256, 256, sc=0.1)), vec(), (vec() + 1, vec()), pw(r(dffn, 256, sc=0.1)), vec(dffn), pw(r(256, dffn, sc=0.05)), vec(),
                    (vec() + 1, vec()), pw(r(256, 256, sc=0.1)), vec(), pw(r(256, 256, sc=0.1)), vec(), r(4, 256, sc=0.1), vec(4), ref)
            o, ro = ops.decoder_tail(*args)
            o, ro = ops.decoder_tail(*args)
            torch.cuda.synchronize()
            outs[f"{dtn}.M{M}.ffn{dffn}.out"], outs[f"{dtn}.M{M}.ffn{dffn}.ref"] = o.cpu(), ro.cpu()
            if dffn == 1024:                      # moy_decoder_mid on the same row counts (n_oa 288 = three levels, 384 = four)
                for n_oa in (288, 384):
                    woa = torch.zeros(max(256, n_oa), 256)
                    woa[:n_oa] = (torch.rand(n_oa, 256, generator=g) - 0.5) * 0.1
                    margs = (samp, e1, r(M, 256).to(dt), pw(r(256, 256, sc=0.1)), vec(), (vec() + 1, vec()), pw(woa.cuda()), vec(n_oa), n_oa)
                    PK = os.environ.get("W_PACKED", "0") == "1"
                    if PK: margs = margs[:3] + (ops.pack_mfma_a(margs[3]),) + margs[4:6] + (ops.pack_mfma_a(margs[6]),) + margs[7:]
                    em, oa = ops.decoder_mid(*margs, packed=PK)
                    em, oa = ops.decoder_mid(*margs, packed=PK)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10): ops.decoder_mid(*margs, packed=PK)
                    torch.cuda.synchronize()
                    if M == 86400: print(f"decoder_mid {dtn} M={M} n_oa={n_oa}: {(time.perf_counter() - t0) / 10 * 1e6:.1f} us per launch")
                    outs[f"{dtn}.M{M}.mid{n_oa}.e1"], outs[f"{dtn}.M{M}.mid{n_oa}.offaw"] = em.cpu(), oa.cpu()
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save"); ap.add_argument("--compare", nargs=2)
    a = ap.parse_args()
    if a.save:
        torch.save(run(), a.save); print("saved", a.save, os.environ.get("MOYOLO_LIB", "(default lib)")); return
    x, y = torch.load(a.compare[0]), torch.load(a.compare[1])
    bad = 0
    for k in x:
        same = torch.equal(x[k], y[k])
        print(f"{k:30s} {'bit-identical' if same else 'DIFFERENT max |d| %.3e' % float((x[k].float() - y[k].float()).abs().max())}")
        bad += not same
    print("ALL BIT-IDENTICAL" if not bad else f"{bad} differ"); sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
