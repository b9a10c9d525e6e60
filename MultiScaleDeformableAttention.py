"""Import shim with the name of the reference's native extension (MOTR/models/ops/setup.py:55,
src/vision.cpp:13-16) so that `import MultiScaleDeformableAttention as MSDA`
(MOTR/models/ops/functions/ms_deform_attn_func.py:21) resolves to libmoyolo.so."""
from mo_yolo_amd.ops import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
