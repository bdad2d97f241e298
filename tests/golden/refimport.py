"""Import the reference's codec classes in THIS container only (never on the GPU box).

The reference tree lives read-only under /root/reference.  Its only missing import on
the hot path is ``pytorch_msssim`` (constructed, never called: common_model.py:29),
which is replaced by an inert placeholder module.  ``MLCodec_CXX`` is served by the
build of the reference's own ops.cpp under oracle/_ref (see oracle/Makefile).
Nothing from the reference is copied into the repo; only small numeric fixtures
produced through it are committed under tests/golden/.
"""
import os
import sys
import types

REF = "/root/reference"


def load(with_cxx: bool = False):
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present (expected only in the build container)")
    if "pytorch_msssim" not in sys.modules:
        m = types.ModuleType("pytorch_msssim")
        m.MS_SSIM = lambda **kw: None
        sys.modules["pytorch_msssim"] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if with_cxx:
        here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        d = os.path.join(here, "oracle", "_ref")
        if d not in sys.path:
            sys.path.insert(0, d)
        import MLCodec_CXX  # built from /root/reference/DCVC_HEM/src/cpp/ops/ops.cpp

        sys.modules["DCVC_HEM.src.entropy_models.MLCodec_CXX"] = MLCodec_CXX
    from DCVC_HEM.src.models.video_model import DMC
    from DCVC_HEM.src.models.image_model import IntraNoAR

    return DMC, IntraNoAR
