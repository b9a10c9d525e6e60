#!/usr/bin/env python3
"""Micro-benchmark of moy_decoder_tail / moy_mlp_head at M = frames*300 rows (BT_B frames): us per call vs d_ffn."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mo_yolo_amd import ops
dev, dt = "cuda", torch.bfloat16
M = int(os.environ.get("BT_B", 96)) * 300
r = lambda *s, sc=1.0: ((torch.rand(*s, device=dev) - 0.5) * sc)
pw = lambda w: ops.pad_weight(w, dt)
samp, e1 = r(M, 256).to(dt), r(M, 256).to(dt)
vec = lambda n=256: r(n, sc=0.2)
ref = torch.rand(M, 4, device=dev)
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1_.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1_) / reps * 1e3
for dffn in (256, 512, 1024, 2048):
    Wp, W1, W2, B0, B1 = pw(r(256, 256, sc=0.1)), pw(r(dffn, 256, sc=0.1)), pw(r(256, dffn, sc=0.05)), pw(r(256, 256, sc=0.1)), pw(r(256, 256, sc=0.1))
    args = (samp, e1, Wp, vec(), (vec() + 1, vec()), W1, vec(dffn), W2, vec(), (vec() + 1, vec()), B0, vec(), B1, vec(), r(4, 256, sc=0.1), vec(4), ref)
    print(f"decoder_tail d_ffn {dffn:5d}: {timeit(lambda: ops.decoder_tail(*args)):7.1f} us")
B0, B1 = pw(r(256, 256, sc=0.1)), pw(r(256, 256, sc=0.1))
w2, c2 = r(4, 256, sc=0.1), vec(4)
print(f"mlp_head: {timeit(lambda: ops.mlp_head(samp, B0, vec(), B1, vec(), w2, c2, mode=1, aux=ref)):7.1f} us")
