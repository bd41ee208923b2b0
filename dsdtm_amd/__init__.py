"""dsdtm_amd — MI355X-native sparse photometric alignment (the DSDTM hot path).

Host-side mirror of DSDTM::Sprase_ImgAlign / DSDTM::Feature_Alignment over the C ABI in
include/dsdtm_amd.h; all compute runs in hand-written HIP kernels for gfx950
(dsdtm_amd/csrc). Importing this package does not load the shared library; the first use
does, and fails loudly when it has not been built.
"""
from .frame import Camera, Config, Frame  # noqa: F401
from .sparse_align import Sprase_ImgAlign  # noqa: F401
from .feature_alignment import Feature_Alignment  # noqa: F401
from .feature_detection import Feature_detector  # noqa: F401
from .optimizer import Optimizer  # noqa: F401

__all__ = ["Camera", "Config", "Frame", "Sprase_ImgAlign", "Feature_Alignment", "Feature_detector",
           "Optimizer"]
