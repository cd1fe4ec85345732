"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/sparenet_hip.h declares (no compute calls here), argument validation returns
SN_EINVAL with a message, and the product path refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "sparenet_hip.h")).read()
    return sorted(set(re.findall(
        r"^(?:int|size_t|void|long long|const char \*)\s*(sn_[a-z0-9_]+)\s*\(", hdr, re.M)))


def test_library_exports_every_declared_symbol():
    import sparenet_amd

    lib = sparenet_amd.lib()
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sparenet_hip.h but not exported"
    assert lib.sn_abi_version() == 4


def test_argument_validation_without_gpu():
    import sparenet_amd

    lib = sparenet_amd.lib()
    null = ctypes.c_void_p(0)
    assert lib.sn_chamfer_forward(null, null, 1, 1, 1, null, null, null, null, null) == -22
    assert b"null pointer" in lib.sn_last_error()
    one = ctypes.c_void_p(8)  # never dereferenced: validation fails first
    assert lib.sn_emd_forward(one, one, 1, 1000, ctypes.c_float(0.005), 5, one, one, one,
                              ctypes.c_size_t(1 << 30), null, null) == -22
    assert b"multiple of 1024" in lib.sn_last_error()
    assert lib.sn_expansion_forward(one, 1, 768, 384, ctypes.c_float(1.5), one, one, one, one,
                                    ctypes.c_size_t(1 << 20), null) == -22
    assert b"power of two" in lib.sn_last_error()
    assert lib.sn_mds(one, 1, 10, 20, one, one, null, ctypes.c_size_t(0), null) == -22
    ctl = 4 * (32 + 2 * 32 * 1024) + 8 * (16 + 64 * 64)   # persistent auction: barrier counters, note blocks + diag words
    # 14 word arrays + 2 arrays of 8-byte entries ({index, rank} lists, {price, index} stream) + 3 of 16-byte entries
    # (bid records {increment, next, index, -}, target records, matrix-core operands) + ...
    assert lib.sn_emd_workspace_bytes(32, 16384) == (14 * 32 * 16384 * 4 + 2 * 32 * 16384 * 8 + 2 * 32 * 256 * 4
                                                      + 3 * 32 * 16384 * 16 + 2 * 32 * 4096 * 4 + 2 * 768
                                                      + 32 * 1024 * 32 + 256 + ctl)   # ... + the far-bidder counters


def test_no_cpu_fallback_anywhere():
    from sparenet_amd import SparenetHipError
    from sparenet_amd.cuda.emd.emd_module import emdModule
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyModule
    from sparenet_amd.cuda.MDS.MDS_module import gather_operation, minimum_density_sample
    from sparenet_amd.cuda.p2i_op import p2i
    from sparenet_amd.cuda.gridding import Gridding, GriddingReverse
    from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSampling

    x = torch.rand(1, 1024, 3)
    with pytest.raises((SparenetHipError, RuntimeError)):
        emdModule()(x, x, 0.005, 2)
    with pytest.raises((SparenetHipError, RuntimeError)):
        expansionPenaltyModule()(x, 512, 1.5)
    with pytest.raises((SparenetHipError, RuntimeError)):
        minimum_density_sample(x, 16, torch.ones(1))
    with pytest.raises((SparenetHipError, RuntimeError)):
        gather_operation(torch.rand(1, 3, 8), torch.zeros(1, 4, dtype=torch.int32))
    with pytest.raises((SparenetHipError, RuntimeError)):
        p2i(torch.rand(4, 2), torch.rand(4, 1), torch.zeros(4, dtype=torch.int32),
            torch.zeros(1, 1, 8, 8), 2.0, "cos", "max")
    with pytest.raises((SparenetHipError, RuntimeError)):
        Gridding(8)(torch.rand(1, 16, 3) * 0.5)
    with pytest.raises((SparenetHipError, RuntimeError)):
        GriddingReverse(4)(torch.rand(1, 4, 4, 4))
    with pytest.raises((SparenetHipError, RuntimeError)):
        CubicFeatureSampling()(torch.rand(1, 8, 3), torch.rand(1, 2, 4, 4, 4))


def test_product_package_never_imports_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "sparenet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M) or "liboracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def _split_top_level(argtext):
    """Split a parenthesised argument text at top-level commas."""
    parts, depth, cur = [], 0, ""
    for ch in argtext:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def _call_sites(src, name):
    """Argument counts of every `name(` call in a Python source text."""
    counts = []
    for m in re.finditer(re.escape(name) + r"\s*\(", src):
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += src[i] in "([{"
            depth -= src[i] in ")]}"
            i += 1
        counts.append(len(_split_top_level(src[m.end():i - 1])))
    return counts


def test_every_python_call_passes_the_declared_number_of_arguments():
    """ctypes does not check argument counts: a parameter added to the C entry point and forgotten at a call site
    shifts every later argument silently.  Every `sn_*(...)` call in the package, bench.py and tools/ is counted
    against the prototype in include/sparenet_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "sparenet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"^(?:int|size_t|void|long long|const char \*)\s*(sn_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr,
                         re.M | re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_top_level(args))
    assert len(protos) >= 40
    checked = 0
    for base in ("sparenet_amd", "tools", "."):
        root = os.path.join(ROOT, base)
        for dirpath, _, files in (os.walk(root) if base != "." else [(root, [], os.listdir(root))]):
            for f in files:
                if not f.endswith(".py"):
                    continue
                src = open(os.path.join(dirpath, f)).read()
                for name, want in protos.items():
                    for got in _call_sites(src, "." + name):
                        # attribute access like `.sn_x.restype = ...` is not a call and is not matched (needs "(")
                        assert got == want, f"{os.path.join(dirpath, f)}: {name} called with {got} arguments, declared {want}"
                        checked += 1
    assert checked >= 40
