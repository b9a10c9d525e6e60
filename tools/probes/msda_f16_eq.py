import os, sys, subprocess, hashlib
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/mo_yolo_amd") else os.getcwd())
if len(sys.argv) > 1:
    import torch
    from mo_yolo_amd import ops
    torch.manual_seed(3)
    dt = torch.float16
    B, S, Lq = 8, 13566, 300
    shapes = [(76, 136), (38, 68), (19, 34), (10, 17)]
    shapes = shapes[:3]; S = sum(h * w for h, w in shapes)
    vp = (torch.rand(6 * 8, B * S, 32, device="cuda") - 0.5).to(dt)
    offaw = (torch.rand(B * Lq, 288, device="cuda") - 0.5) * 4
    ref = torch.rand(B * Lq, 4, device="cuda") * 0.5 + 0.25
    y = ops.msda_fused(vp[:8], B, S, shapes, offaw, ref, Lq, head_planes=True)
    torch.cuda.synchronize()
    torch.save(y.cpu(), sys.argv[2])
    print("HASH", hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest(), tuple(y.shape), y.dtype)
else:
    for lib in ("libmoyolo_base.so", "libmoyolo.so"):
        env = dict(os.environ, MOYOLO_LIB=os.path.join(os.getcwd(), "mo_yolo_amd", lib))
        r = subprocess.run([sys.executable, __file__, "child", "/tmp/" + lib + ".pt"], env=env, capture_output=True, text=True)
        print(lib, [l for l in r.stdout.splitlines() if l.startswith("HASH")] or r.stderr[-500:])
    import torch
    a, b = torch.load("/tmp/libmoyolo_base.so.pt").float(), torch.load("/tmp/libmoyolo.so.pt").float()
    d = (a - b).abs()
    print("elements differing:", int((d > 0).sum()), "of", d.numel(), " max abs diff", float(d.max()), " max |value|", float(a.abs().max()),
          " max diff in fp16 ulps of the value:", float((d / (a.abs().clamp_min(6e-5) * 2 ** -10)).max()))
