"""Minimum density sampling and point gathering -- host-side mirror of
cuda/MDS/MDS_module.py (MinimumDensitySampling :7-38, minimum_density_sample :41,
GatherOperation :44-75, gather_operation :78), backed by sn_mds / sn_gather_forward /
sn_gather_backward (include/sparenet_hip.h).

    minimum_density_sample(xyz [B,N,3], npoint, mean_mst_length [B]) -> idx [B,npoint] int32
        greedy sampling that repeatedly takes the point of lowest accumulated Gaussian
        density; mean_mst_length comes from the expansion penalty module.  Not differentiable.
    gather_operation(features [B,C,N], idx [B,npoint]) -> [B,C,npoint], differentiable in
        features.
"""
import ctypes

import torch
from torch.autograd import Function

from sparenet_amd import _lib


class MinimumDensitySampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint, mean_mst_length):
        if xyz.dim() != 3 or xyz.size(2) != 3:
            raise ValueError("minimum_density_sample: xyz must be [B, N, 3]")
        xyz = xyz.contiguous().float()
        mean_mst_length = mean_mst_length.contiguous().float()
        b, n, _ = xyz.shape
        idx = torch.empty(b, npoint, device=xyz.device, dtype=torch.int32)
        with torch.cuda.device_of(xyz):
            nbytes = _lib.lib().sn_mds_workspace_bytes(b, n)
            ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=xyz.device)
            code = _lib.lib().sn_mds(
                _lib.fptr(xyz, "xyz"), b, n, int(npoint),
                _lib.fptr(mean_mst_length, "mean_mst_length"), _lib.iptr(idx, "idx"),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), _lib.stream_of(xyz))
        _lib.check(code, "sn_mds")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_idx=None):
        return None, None, None


minimum_density_sample = MinimumDensitySampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        features = features.contiguous().float()
        idx = idx.contiguous()
        b, c, n = features.size()
        m = idx.size(1)
        ctx.for_backwards = (idx, c, n)
        out = torch.empty(b, c, m, device=features.device)
        with torch.cuda.device_of(features):
            code = _lib.lib().sn_gather_forward(
                _lib.fptr(features, "features"), _lib.iptr(idx, "idx"), b, c, n, m,
                _lib.fptr(out, "out"), _lib.stream_of(features))
        _lib.check(code, "sn_gather_forward")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, c, n = ctx.for_backwards
        grad_out = grad_out.contiguous().float()
        b, _, m = grad_out.shape
        grad_features = torch.empty(b, c, n, device=grad_out.device)
        with torch.cuda.device_of(grad_out):
            code = _lib.lib().sn_gather_backward(
                _lib.fptr(grad_out, "grad_out"), _lib.iptr(idx, "idx"), b, c, n, m,
                _lib.fptr(grad_features, "grad_features"), _lib.stream_of(grad_out))
        _lib.check(code, "sn_gather_backward")
        return grad_features, None


gather_operation = GatherOperation.apply
