"""cudaraytracer_amd — MI355X (gfx950) drop-in for the list-render path of
chocorone/CudaRaytracer (render kernel + HitableList traversal + triangle test +
LambertShade).

The product is the C-ABI library ``libcrt_hip.so`` (include/crt.h, sources in csrc/).
This package is the thin host-side mirror of the reference's host interface on top of it
(``api``), the synthetic scene generators (``scenes``) and the multi-GPU row-band
partition + RCCL gather (``dist``).  There is no CPU compute path: every render call
fails loudly when the HIP library or a HIP device is missing.
"""
from . import _lib  # noqa: F401
from ._lib import CrtError, build_library, library_path  # noqa: F401

__all__ = ["CrtError", "build_library", "library_path", "api", "scenes", "dist"]
