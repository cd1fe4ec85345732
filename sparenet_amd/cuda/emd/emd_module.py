"""EMD approximation module (auction algorithm) -- host-side mirror of
cuda/emd/emd_module.py (emdFunction :30-87, emdModule :90-95).

Input:  xyz1 (prediction), xyz2 (ground truth): [#batch, #points, 3], same size,
        coordinates normalised to [0, 1]; #points a multiple of 1024; #batch <= 512.
        eps balances error rate against convergence speed; iters = auction rounds.
Output: dist [#batch, #points] (sqrt(dist) -> L2 distance), assignment
        [#batch, #points] int32 (index of the matched ground-truth point; an
        approximation, not guaranteed to be a bijection).  Gradient only for xyz1.

Backed by sn_emd_forward / sn_emd_backward (include/sparenet_hip.h); the twelve
scratch tensors the reference allocates per call (:43-54) are one workspace here.
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function

from sparenet_amd import _lib


def emd_forward_raw(xyz1, xyz2, eps, iters, stats=None, return_workspace=False):
    """C-ABI call on contiguous fp32 CUDA tensors; returns (dist, assignment).
    stats: optional int64[2] CUDA tensor accumulating (effective pairs, active iterations)."""
    batchsize, n, _ = xyz1.size()
    dev = xyz1.device
    dist = torch.empty(batchsize, n, device=dev)
    assignment = torch.empty(batchsize, n, device=dev, dtype=torch.int32)
    with torch.cuda.device_of(xyz1):
        nbytes = _lib.lib().sn_emd_workspace_bytes(batchsize, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        sp = ctypes.c_void_p(stats.data_ptr()) if stats is not None else ctypes.c_void_p(0)
        code = _lib.lib().sn_emd_forward(
            _lib.fptr(xyz1, "xyz1"), _lib.fptr(xyz2, "xyz2"), batchsize, n, _lib.cfloat(eps),
            int(iters), _lib.fptr(dist, "dist"), _lib.iptr(assignment, "assignment"),
            ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), sp, _lib.stream_of(xyz1))
    _lib.check(code, "sn_emd_forward")
    if return_workspace:
        return dist, assignment, ws
    return dist, assignment


class emdFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()

        assert n == m
        assert xyz1.size()[0] == xyz2.size()[0]
        assert n % 1024 == 0
        assert batchsize <= 512

        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        # (an EARLIER launch's team time-out fails this call inside the library -- sn_emd_forward checks the device's
        # sticky word first; a time-out of THIS launch surfaces at the next op or at sparenet_amd.loss_item)
        dist, assignment = emd_forward_raw(xyz1, xyz2, eps, iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        graddist = graddist.contiguous().float()
        gradxyz1 = torch.empty_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        b, n, _ = xyz1.shape
        with torch.cuda.device_of(xyz1):
            code = _lib.lib().sn_emd_backward(
                _lib.fptr(xyz1, "xyz1"), _lib.fptr(xyz2, "xyz2"), _lib.fptr(graddist, "graddist"),
                _lib.iptr(assignment, "assignment"), b, n, _lib.fptr(gradxyz1, "gradxyz1"),
                _lib.stream_of(xyz1))
        _lib.check(code, "sn_emd_backward")
        return gradxyz1, gradxyz2, None, None


class emdModule(nn.Module):
    def __init__(self):
        super(emdModule, self).__init__()

    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)
