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


# The C ABI this Python side was written against (include/sparenet_hip.h: SN_ABI_VERSION).  The library is built
# separately and is not tracked: a stale .so next to newer Python (or the reverse) would make ctypes pass shifted
# arguments -- memory corruption instead of an error -- so lib() refuses any other version.
EXPECTED_ABI = 4

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
           "long long": ctypes.c_longlong, "long": ctypes.c_long, "unsigned": ctypes.c_uint, "void": None}


def _prototypes():
    """{name: (restype, [argtypes])} parsed from include/sparenet_hip.h, or {} when the header is not shipped
    next to the package (the version check above still applies)."""
    import re

    hdr = os.path.join(os.path.dirname(_HERE), "include", "sparenet_hip.h")
    if not os.path.isfile(hdr):
        return {}
    txt = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"^(int|size_t|void|long long|const char \*)\s*(sn_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", txt,
                         re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(ctypes.c_char_p if a.startswith("const char") else ctypes.c_void_p)
                else:
                    base = a.rsplit(" ", 1)[0].replace("const ", "").strip()
                    types.append(_CTYPES[base])
        out[name] = (ctypes.c_char_p if "char" in ret else _CTYPES[ret], types)
    return out


def lib():
    """Load (once) and return the HIP library; raises if it is not built or was built for another ABI."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SparenetHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C sparenet_amd/csrc`. sparenet_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        got = L.sn_abi_version()
        if got != EXPECTED_ABI:
            raise SparenetHipError(
                f"{LIB_PATH} implements C ABI version {got}, this Python side needs {EXPECTED_ABI}: rebuild the "
                "library (`make -C sparenet_amd/csrc`)")
        L.sn_last_error.restype = ctypes.c_char_p
        L.sn_build_id.restype = ctypes.c_char_p
        for name, (ret, args) in _prototypes().items():
            fn = getattr(L, name, None)
            if fn is not None:      # a missing export is test_abi's finding, not a load-time failure
                fn.restype = ret
                fn.argtypes = args
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().sn_last_error().decode("utf-8", "replace")
        raise SparenetHipError(f"{what} failed (code {code}): {msg}")


def device_check(what="sparenet_amd"):
    """Raise if a bounded wait inside an EARLIER multi-workgroup launch on the current device gave up (the persistent EMD
    auction's team barriers, the density sampler's teams; that call's outputs are NaN / -1).  Reads one word of pinned
    host memory (sn_device_status): no synchronisation -- meaningful for work that has FINISHED, so the wrappers call
    it on entry (an earlier step's failure surfaces at the next op) and `loss_item` calls it where the host has just
    waited for the loss."""
    check(lib().sn_device_status(), what)


def ptr(t, dtype, name):
    """Device pointer of a contiguous CUDA tensor (validated)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise SparenetHipError(
            f"{name}: expected a CUDA (ROCm) tensor, got device {t.device}. sparenet_amd runs on "
            "MI355X only; there is no CPU path for this op (the reference has none either; only ChamferDistance "
            "accepts CPU tensors, as in the reference).")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def hptr(t, dtype, name):
    """HOST pointer of a contiguous CPU tensor (validated) -- only the Chamfer host entry points take these."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.is_cuda:
        raise SparenetHipError(f"{name}: expected a CPU tensor, got device {t.device}")
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
