"""sparenet_amd -- MI355X (gfx950) native SpareNet loss/render hot path.

Host-side mirror of the reference's operator API lives in sparenet_amd.cuda.*
and sparenet_amd.utils.p2i_utils (same module names, class names and argument
meaning as /root/reference/cuda/* and utils/p2i_utils.py); every op calls the
hand-written HIP kernels in libsparenet_hip.so through the C ABI declared in
include/sparenet_hip.h.  There is no CPU or eager-PyTorch fallback.
"""
from ._lib import LIB_PATH, SparenetHipError, device_check, lib  # noqa: F401

__version__ = "0.1.0"

_REFERENCE_OP_PACKAGES = ("chamfer_distance", "chamfer_dist", "emd", "expansion_penalty", "MDS", "p2i_op",
                          "gridding", "gridding_loss", "cubic_feature_sampling")


def loss_item(loss):
    """`loss.item()` that cannot hand back the number of a failed step: the host waits for the GPU (as `.item()` always
    does), then the device's sticky error word is read (sn_device_status) -- if a team barrier of the persistent EMD
    auction or of the density sampler timed out in any launch up to here (a shared device, a debugger holding a compute
    unit), its outputs were NaN / -1 and this raises SparenetHipError instead of returning NaN to the training loop.
    Where the reference's runners log `_loss.item()` every step (runners/sparenet_runner.py:113-116), log
    `sparenet_amd.loss_item(_loss)` -- before `optimizer.step()` if a failed step must not touch the weights."""
    value = loss.item()
    device_check("loss_item")
    return value


def alias_reference_modules():
    """Make the reference's import lines resolve to this package without touching its checkout:
    `from cuda.emd.emd_module import emdModule` (runners/sparenet_runner.py:9), `import cuda.MDS.MDS_module`
    (models/sparenet_generator.py:8), `from cuda.p2i_op import p2i`, `from utils.p2i_utils import
    ComputeDepthMaps` (utils/model_init.py:9) ... -- INTEGRATION.md section 2.  Call it once at start-up,
    before the runners are imported."""
    import importlib
    import sys

    amd_cuda = importlib.import_module("sparenet_amd.cuda")
    sys.modules["cuda"] = amd_cuda
    for name in _REFERENCE_OP_PACKAGES:
        mod = importlib.import_module(f"sparenet_amd.cuda.{name}")
        sys.modules[f"cuda.{name}"] = mod
        for sub in ("emd_module", "expansion_penalty_module", "MDS_module", "chamfer_distance"):
            try:
                sys.modules[f"cuda.{name}.{sub}"] = importlib.import_module(f"sparenet_amd.cuda.{name}.{sub}")
            except ModuleNotFoundError:
                pass
    amd_p2i = importlib.import_module("sparenet_amd.utils.p2i_utils")
    sys.modules["utils.p2i_utils"] = amd_p2i
    utils_pkg = sys.modules.get("utils")
    if utils_pkg is not None:                 # the reference's own `utils` package, already imported
        setattr(utils_pkg, "p2i_utils", amd_p2i)
    return amd_cuda
