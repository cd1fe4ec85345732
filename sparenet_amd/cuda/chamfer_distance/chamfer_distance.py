"""Chamfer distance -- host-side mirror of cuda/chamfer_distance/chamfer_distance.py.

Same public names as the reference (ChamferDistanceFunction :19-61,
ChamferDistance :64-66, ChamferDistanceMean :69-72, module global `cd` :8-15),
backed by sn_chamfer_forward / sn_chamfer_backward (include/sparenet_hip.h).
CPU tensors take the branch the reference takes for them (:31-32, :53-54 -> cd.forward /
cd.backward): the library's own host implementation (sn_chamfer_*_host, csrc/chamfer_host.hip),
bit-equal to the reference's CPU code.  It is the ONLY op with a host path, because it is the only
one the reference gives one; CUDA tensors never take it.
"""
import ctypes

import torch

from sparenet_amd import _lib


class _CdBinding:
    """Stand-in for the reference's JIT-built `cd` extension module
    (chamfer_distance.cpp:182-188): caller-allocated outputs, GPU entry points."""

    # below this many pairs per cloud the all-pairs kernel wins over sort + pruned search
    SORTED_MIN_PAIRS = 1 << 22

    @staticmethod
    def forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2):
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        if n * m >= _CdBinding.SORTED_MIN_PAIRS and b * max(n, m) < (1 << 26):
            return _CdBinding.forward_sorted_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)
        with torch.cuda.device_of(xyz1):
            code = _lib.lib().sn_chamfer_forward(
                _lib.fptr(xyz1, "xyz1"), _lib.fptr(xyz2, "xyz2"), b, n, m,
                _lib.fptr(dist1, "dist1"), _lib.iptr(idx1, "idx1"),
                _lib.fptr(dist2, "dist2"), _lib.iptr(idx2, "idx2"), _lib.stream_of(xyz1))
        _lib.check(code, "sn_chamfer_forward")

    @staticmethod
    def forward_sorted_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2):
        """sn_chamfer_forward_sorted: identical outputs through a spatially pruned search."""
        import ctypes
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        with torch.cuda.device_of(xyz1):
            nbytes = _lib.lib().sn_chamfer_workspace_bytes(b, n, m)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz1.device)
            code = _lib.lib().sn_chamfer_forward_sorted(
                _lib.fptr(xyz1, "xyz1"), _lib.fptr(xyz2, "xyz2"), b, n, m,
                _lib.fptr(dist1, "dist1"), _lib.iptr(idx1, "idx1"),
                _lib.fptr(dist2, "dist2"), _lib.iptr(idx2, "idx2"),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), _lib.stream_of(xyz1))
        _lib.check(code, "sn_chamfer_forward_sorted")

    @staticmethod
    def backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        with torch.cuda.device_of(xyz1):
            nbytes = _lib.lib().sn_chamfer_backward_workspace_bytes(b, n, m)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz1.device)   # inverse neighbour lists
            code = _lib.lib().sn_chamfer_backward(
                _lib.fptr(xyz1, "xyz1"), _lib.fptr(xyz2, "xyz2"),
                _lib.fptr(graddist1, "graddist1"), _lib.fptr(graddist2, "graddist2"),
                _lib.iptr(idx1, "idx1"), _lib.iptr(idx2, "idx2"), b, n, m,
                _lib.fptr(gradxyz1, "gradxyz1"), _lib.fptr(gradxyz2, "gradxyz2"),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), _lib.stream_of(xyz1))
        _lib.check(code, "sn_chamfer_backward")

    @staticmethod
    def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        """cd.forward (chamfer_distance.cpp:91-112): host tensors, caller-allocated outputs."""
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        code = _lib.lib().sn_chamfer_forward_host(
            _lib.hptr(xyz1, torch.float32, "xyz1"), _lib.hptr(xyz2, torch.float32, "xyz2"), b, n, m,
            _lib.hptr(dist1, torch.float32, "dist1"), _lib.hptr(idx1, torch.int32, "idx1"),
            _lib.hptr(dist2, torch.float32, "dist2"), _lib.hptr(idx2, torch.int32, "idx2"), 0)
        _lib.check(code, "sn_chamfer_forward_host")

    @staticmethod
    def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        """cd.backward (chamfer_distance.cpp:114-180): host tensors; the gradients are fully overwritten."""
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        code = _lib.lib().sn_chamfer_backward_host(
            _lib.hptr(xyz1, torch.float32, "xyz1"), _lib.hptr(xyz2, torch.float32, "xyz2"),
            _lib.hptr(graddist1, torch.float32, "graddist1"), _lib.hptr(graddist2, torch.float32, "graddist2"),
            _lib.hptr(idx1, torch.int32, "idx1"), _lib.hptr(idx2, torch.int32, "idx2"), b, n, m,
            _lib.hptr(gradxyz1, torch.float32, "gradxyz1"), _lib.hptr(gradxyz2, torch.float32, "gradxyz2"), 0)
        _lib.check(code, "sn_chamfer_backward_host")


cd = _CdBinding()


def _check_pair(xyz1, xyz2):
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.size(2) != 3 or xyz2.size(2) != 3:
        raise ValueError("ChamferDistance expects xyz1 [B,N,3] and xyz2 [B,M,3]")
    if xyz1.size(0) != xyz2.size(0):
        raise ValueError("ChamferDistance: batch sizes differ")
    if xyz1.size(1) == 0 or xyz2.size(1) == 0 or xyz1.size(0) == 0:
        raise ValueError("ChamferDistance: empty point cloud (undefined in the reference too)")


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        _check_pair(xyz1, xyz2)
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        dev = xyz1.device
        if xyz2.device != dev:
            raise ValueError("ChamferDistance: xyz1 and xyz2 are on different devices")
        dist1 = torch.empty(batchsize, n, device=dev)
        dist2 = torch.empty(batchsize, m, device=dev)
        idx1 = torch.empty(batchsize, n, dtype=torch.int, device=dev)
        idx2 = torch.empty(batchsize, m, dtype=torch.int, device=dev)
        if not xyz1.is_cuda:      # the reference's branch for CPU tensors (chamfer_distance.py:31-32)
            cd.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        else:
            cd.forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, graddist1, graddist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous().float()
        graddist2 = graddist2.contiguous().float()
        gradxyz1 = torch.empty_like(xyz1)
        gradxyz2 = torch.empty_like(xyz2)
        if not graddist1.is_cuda:  # chamfer_distance.py:53-54
            cd.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        else:
            cd.backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        return gradxyz1, gradxyz2


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)


class ChamferDistanceMean(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        dist1, dist2 = ChamferDistanceFunction.apply(xyz1, xyz2)
        return (torch.mean(dist1)) + (torch.mean(dist2))
