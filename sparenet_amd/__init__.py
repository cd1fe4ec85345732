"""sparenet_amd -- MI355X (gfx950) native SpareNet loss/render hot path.

Host-side mirror of the reference's operator API lives in sparenet_amd.cuda.*
and sparenet_amd.utils.p2i_utils (same module names, class names and argument
meaning as /root/reference/cuda/* and utils/p2i_utils.py); every op calls the
hand-written HIP kernels in libsparenet_hip.so through the C ABI declared in
include/sparenet_hip.h.  There is no CPU or eager-PyTorch fallback.
"""
from ._lib import LIB_PATH, SparenetHipError, lib  # noqa: F401

__version__ = "0.1.0"
