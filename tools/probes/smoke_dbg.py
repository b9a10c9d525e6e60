import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorName.restype = ctypes.c_char_p
def last(tag):
    e = hip.hipGetLastError()
    print(tag, e, hip.hipGetErrorName(e))
from mo_yolo_amd.engine import TrackEngine
from mo_yolo_amd.fixtures import fixture
from mo_yolo_amd.synth import SyntheticSequence
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "build":
    import __graft_entry__ as g
    g.build()
last("after imports")
cfg, arch, sd = fixture("tiny")
fr = SyntheticSequence(0, cfg["H"], cfg["W"], cfg["style"]).frames(0, 2)
last("before engine")
eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=torch.float32)
last("after engine")
x = torch.from_numpy(fr).to("cuda:0")
last("after h2d")
try:
    out = eng.forward(x)
    torch.cuda.synchronize()
    print("forward ok")
except Exception as e:
    print("forward failed", e)
    last("after fail")
    out = eng.forward(x)
    torch.cuda.synchronize()
    print("second forward ok")
