"""v3d: MI355X-native kernels for Video-3D-LLM's position-aware video->LLM forward path.

`v3d.ops` wraps the C ABI (include/v3d.h) for torch tensors that already live in HBM.
"""
from ._native import V3DError, lib, LIB_PATH  # noqa: F401
