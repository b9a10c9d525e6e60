"""Import shim for the read-only reference tree (/root/reference).

ONLY used in the build container to GENERATE golden vectors (tests/golden/make_golden.py)
and to cross-check the oracle restatement.  Nothing here travels as reference code: the
GPU box has no /root/reference and no test marked `gpu` imports this module.

Recipe: SURVEY.md Appendix B.  The reference does not import as shipped
(head.py:107-111 -> MOTR/main.py:23-31 -> absent MOTR.datasets; un-built
MultiScaleDeformableAttention; cv2/torchvision absent in this image), so a meta_path
finder serves MagicMock modules for exactly those prefixes.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("MOY_REFERENCE_ROOT", "/root/reference")
_STUB_PREFIXES = ("cv2", "torchvision", "MultiScaleDeformableAttention", "MOTR.datasets",
                  "seaborn", "thop", "cpuinfo", "pycocotools", "motmetrics")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name == "__version__":
            return "0.15.0"
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        m = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        for p in _STUB_PREFIXES:
            if fullname == p or fullname.startswith(p + "."):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "ultralytics"))


def install():
    """Make `import ultralytics` / `import MOTR` resolve to the reference tree."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF_ROOT)
    sys.argv = sys.argv[:1]  # head.py:110-111 parses sys.argv inside the module ctor
    os.environ.setdefault("YOLO_OFFLINE", "1")
    import cv2  # noqa: the stub

    class _VW:  # predictor.py:331 uses isinstance(..., cv2.VideoWriter)
        pass
    cv2.VideoWriter = _VW
    _installed = True


def build_tracking_model(depth=0.33, width=0.50, nc=1, yaml_name="yolo_track.yaml"):
    """TrackingModel from the reference yaml at the given scale (s-scale = 0.33/0.50)."""
    install()
    import torch
    from ultralytics.nn.tasks import TrackingModel, yaml_model_load
    cfg = yaml_model_load(os.path.join(REF_ROOT, "ultralytics/cfg/models/v8", yaml_name))
    cfg["nc"] = nc
    cfg["depth_multiple"] = depth
    cfg["width_multiple"] = width
    cfg.pop("scales", None)
    with torch.no_grad():
        model = TrackingModel(cfg, ch=3, nc=nc, verbose=False).eval()
    return model
