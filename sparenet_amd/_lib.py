"""ctypes binding of libsparenet_hip.so (the C ABI declared in include/sparenet_hip.h).

There is deliberately NO fallback: if the HIP library is missing, or a tensor is
not a contiguous CUDA(ROCm) tensor of the right dtype, these helpers raise.
PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsparenet_hip.so")
_lib = None

SN_EINVAL = -22


class SparenetHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SparenetHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C sparenet_amd/csrc`. sparenet_amd has no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.sn_last_error.restype = ctypes.c_char_p
        _lib.sn_build_id.restype = ctypes.c_char_p
        _lib.sn_prof_read.restype = ctypes.c_longlong
        _lib.sn_prof_enable.restype = None
        _lib.sn_prof_reset.restype = None
        for name in ("sn_emd_workspace_bytes", "sn_emd_diag_offset", "sn_p2i_max_workspace_bytes",
                     "sn_depthmaps_workspace_bytes", "sn_expansion_workspace_bytes", "sn_mds_workspace_bytes", "sn_p2i_max_backward_workspace_bytes",
                     "sn_p2i_max_multi_workspace_bytes",
                     "sn_p2i_max_backward_multi_workspace_bytes", "sn_chamfer_workspace_bytes", "sn_chamfer_backward_workspace_bytes", "sn_p2i_f64_workspace_bytes",
                     "sn_graph_feature_backward_workspace_bytes", "sn_knn_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_size_t
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().sn_last_error().decode("utf-8", "replace")
        raise SparenetHipError(f"{what} failed (code {code}): {msg}")


def ptr(t, dtype, name):
    """Device pointer of a contiguous CUDA tensor (validated)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise SparenetHipError(
            f"{name}: expected a CUDA (ROCm) tensor, got device {t.device}. sparenet_amd runs on "
            "MI355X only; there is no CPU path (the reference's own CPU Chamfer lives in "
            "/root/reference/cuda/chamfer_distance/chamfer_distance.cpp).")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def fptr(t, name):
    return ptr(t, torch.float32, name)


def iptr(t, name):
    return ptr(t, torch.int32, name)


def dptr(t, name):
    return ptr(t, torch.float64, name)


def stream_of(t):
    """Current HIP stream of the tensor's device, as void*."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def cfloat(x):
    return ctypes.c_float(float(x))
