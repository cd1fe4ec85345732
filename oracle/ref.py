"""oracle/ref.py -- loader for oracle/_ref/cd_ref.so (TEST INFRASTRUCTURE ONLY).

cd_ref.so is the reference's own Chamfer CPU path
(/root/reference/cuda/chamfer_distance/chamfer_distance.cpp, compiled unmodified
by `make -C oracle ref`).  Its pybind module exports forward/backward (CPU) and
forward_cuda/backward_cuda; the latter reference two CUDA launcher symbols that
do not exist here, so the library is dlopen'ed with RTLD_LAZY and only the CPU
entry points are ever called.
"""
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "cd_ref.so")


def available() -> bool:
    return os.path.isfile(REF_SO)


_mod = None


def load():
    """Return the reference pybind module (forward, backward on CPU tensors)."""
    global _mod
    if _mod is not None:
        return _mod
    if not available():
        raise FileNotFoundError(
            f"{REF_SO} missing: run `make -C oracle ref` where /root/reference exists")
    import torch  # noqa: F401  (libtorch symbols must be loaded first)

    flags = sys.getdlopenflags()
    try:
        sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
        loader = importlib.machinery.ExtensionFileLoader("cd_ref", REF_SO)
        spec = importlib.util.spec_from_file_location("cd_ref", REF_SO, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
    finally:
        sys.setdlopenflags(flags)
    _mod = mod
    return mod


def chamfer_forward(xyz1, xyz2):
    """xyz1 [B,N,3], xyz2 [B,M,3] CPU float tensors -> dist1, dist2, idx1, idx2."""
    import torch

    cd = load()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    dist1 = torch.zeros(b, n)
    dist2 = torch.zeros(b, m)
    idx1 = torch.zeros(b, n, dtype=torch.int)
    idx2 = torch.zeros(b, m, dtype=torch.int)
    cd.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
    return dist1, dist2, idx1, idx2


def chamfer_backward(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    import torch

    cd = load()
    g1 = torch.zeros_like(xyz1)
    g2 = torch.zeros_like(xyz2)
    cd.backward(xyz1.contiguous(), xyz2.contiguous(), g1, g2,
                graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
    return g1, g2
