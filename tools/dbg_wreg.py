import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mo_yolo_amd import _lib as L, ops
DEV = "cuda"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
for dt in (torch.bfloat16, torch.float16):
    for M, N, act in [(70000, 256, "none"), (65537, 512, "silu"), (65537, 512, "none"), (66063, 1536, "none")]:
        K = 256
        x, w = rnd(M, K, seed=11).to(dt), rnd(N, K, seed=12, scale=1 / math.sqrt(K)).to(dt)
        b = rnd(N, seed=13, scale=0.1).to(DEV)
        sc = (rnd(N, seed=14) * 0.2 + 1.0).to(DEV) if act == "silu" else None
        xd, wd = x.to(DEV), ops.pad_weight(w.float().to(DEV), dt)
        kw = dict(shift=b, scale=sc, act=L.ACT_SILU if act == "silu" else L.ACT_NONE)
        out = ops.gemm(xd, wd, N, K, **kw)
        h = M // 2
        two = torch.empty(M, N, device=DEV, dtype=dt)
        ops.gemm(xd[:h], wd, N, K, out=two[:h], **kw)
        ops.gemm(xd[h:], wd, N, K, out=two[h:], **kw)
        torch.cuda.synchronize()
        d = (out.float() - two.float()).abs()
        bad = (out != two)
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print(dt, M, N, act, "mismatch", int(bad.sum()), "maxdiff", float(d.max()), "rows", rows[:8].tolist(), len(rows), "cols", cols[:8].tolist(), len(cols))
        if int(bad.sum()):
            i = bad.nonzero()[0]
            print("  first", i.tolist(), float(out[i[0], i[1]]), float(two[i[0], i[1]]))
