"""Expansion penalty -- host-side mirror of
cuda/expansion_penalty/expansion_penalty_module.py (expansionPenaltyFunction
:23-48, expansionPenaltyModule :51-56).

forward(input [B,n,3], primitive_size, alpha) ->
    (dist [B,n], assignment [B,n] int32, mean_mst_length [B])
GPU tensors only.  Backed by sn_expansion_forward / sn_expansion_backward; the
reference's two [B, n*512] neighbor/cost scratch tensors (:33-34) do not exist.
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function

from sparenet_amd import _lib


class expansionPenaltyFunction(Function):
    @staticmethod
    def forward(ctx, xyz, primitive_size, alpha):
        assert primitive_size <= 512
        batchsize, n, _ = xyz.size()
        assert n % primitive_size == 0
        xyz = xyz.contiguous().float()
        dev = xyz.device
        dist = torch.empty(batchsize, n, device=dev)
        assignment = torch.empty(batchsize, n, device=dev, dtype=torch.int32)
        mean_mst_length = torch.empty(batchsize, device=dev)
        with torch.cuda.device_of(xyz):
            nbytes = _lib.lib().sn_expansion_workspace_bytes(batchsize, n, int(primitive_size))
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
            code = _lib.lib().sn_expansion_forward(
                _lib.fptr(xyz, "xyz"), batchsize, n, int(primitive_size), _lib.cfloat(alpha),
                _lib.fptr(dist, "dist"), _lib.iptr(assignment, "assignment"),
                _lib.fptr(mean_mst_length, "mean_mst_length"), ctypes.c_void_p(ws.data_ptr()),
                ctypes.c_size_t(nbytes), _lib.stream_of(xyz))
        _lib.check(code, "sn_expansion_forward")
        ctx.save_for_backward(xyz, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment, mean_mst_length / (n / primitive_size)

    @staticmethod
    def backward(ctx, grad_dist, grad_idx, grad_mml):
        # only `dist` carries a gradient; mean_mst_length is treated as a constant, as in the
        # reference (expansion_penalty_module.py:42-48)
        xyz, assignment = ctx.saved_tensors
        grad_dist = grad_dist.contiguous().float()
        b, n, _ = xyz.shape
        grad_xyz = torch.empty_like(xyz)
        with torch.cuda.device_of(xyz):
            code = _lib.lib().sn_expansion_backward(
                _lib.fptr(xyz, "xyz"), _lib.fptr(grad_dist, "grad_dist"),
                _lib.iptr(assignment, "assignment"), b, n, _lib.fptr(grad_xyz, "grad_xyz"),
                _lib.stream_of(xyz))
        _lib.check(code, "sn_expansion_backward")
        return grad_xyz, None, None


class expansionPenaltyModule(nn.Module):
    """nn.Module face of expansionPenaltyFunction (the generator instantiates it once and
    calls it per refinement stage, models/sparenet_generator.py:551,559)."""

    def forward(self, input, primitive_size, alpha):
        dist, assignment, mean_mst_length = expansionPenaltyFunction.apply(input, primitive_size, alpha)
        return dist, assignment, mean_mst_length
