"""Probe: phase stamps of the fused decoder tail (MOY_TAIL_ABL=2, wave 0 of the first and the middle block) and its time at the C2 bench size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops
M, dffn, dt = int(os.environ.get("TAIL_M", 86400)), 1024, torch.bfloat16
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: ((torch.rand(*s, generator=g) - 0.5) * sc).cuda()
pw = lambda w: ops.pad_weight(w, dt)
vec = lambda n=256: r(n, sc=0.2)
samp, e1 = r(M, 256).to(dt), r(M, 256).to(dt)
ref = torch.rand(M, 4, generator=g).cuda()
args = [samp, e1, pw(r(256, 256, sc=0.1)), vec(), (vec() + 1, vec()), pw(r(dffn, 256, sc=0.1)), vec(dffn), pw(r(256, dffn, sc=0.05)), vec(),
        (vec() + 1, vec()), pw(r(256, 256, sc=0.1)), vec(), pw(r(256, 256, sc=0.1)), vec(), r(4, 256, sc=0.1), vec(4), ref]
o0, ro0 = ops.decoder_tail(*args); torch.cuda.synchronize()
PACKED = os.environ.get("TAIL_PACKED", "1") != "0"
if PACKED:
    for i in (2, 5, 7, 10, 12): args[i] = ops.pack_mfma_a(args[i])
_tail = ops.decoder_tail
ops.decoder_tail = lambda *a: _tail(*a, packed=PACKED)
o, ro = ops.decoder_tail(*args); torch.cuda.synchronize()
if not os.environ.get("MOY_TAIL_ABL"):
    print("fragment-ordered weights" if PACKED else "row-major weights", "- outputs equal to the row-major call:", torch.equal(o, o0) and torch.equal(ro, ro0))
e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): o, ro = ops.decoder_tail(*args)
e1_.record(); torch.cuda.synchronize()
print(f"decoder_tail M={M}: {e0.elapsed_time(e1_) / 10 * 1e3:.1f} us per launch")
if os.environ.get("MOY_TAIL_ABL") in ("2", "3"):
    d = o.view(torch.int64).flatten()[:32].cpu().tolist()
    w = o.view(torch.int64).flatten()[64:128].view(8, 8)[:, :6].cpu()
    t0 = int(w[:, 0].min())
    print("every wave of block 0 through FFN chunk 1 (clock ticks from the first wave's start): linear1 start | end | relu+put end | barrier out | linear2 end | barrier out")
    for i in range(8):
        print(f"   wave {i}: " + " ".join(f"{int(x) - t0:7d}" for x in w[i]))
    names = ["load samp + barrier", "GEMM output_proj", "bias + residual + LN2 + put + barrier", "GEMM linear1 (4 chunks)", "relu + put + barrier (4)", "GEMM linear2 + barrier (4)",
             "bias + residual + LN3 + put + barrier", "store out (+ out_xp)", "GEMM box 0", "relu + put + barrier", "GEMM box 1", "dots + reduce + sigmoid",
             "  (LN2 epilogue up to its last barrier; row 3 = that barrier)", "  (relu + put of linear1; row 5 = the barrier after it)", "  (linear2 products; row 6 = the barrier after them)", "-"]
    for blk, base in (("first block", 0), ("middle block", 16)):
        v = d[base:base + 16]
        tot = sum(v)
        print(f"{blk}: total {tot} ticks of s_memtime (100 MHz: {tot / 100:.1f} us)")
        for nm, x in zip(names, v):
            print(f"   {nm:42s} {x:8d}  {100 * x / max(tot, 1):5.1f} %")
