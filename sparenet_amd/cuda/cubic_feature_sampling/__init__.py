"""CubicFeatureSampling -- host-side mirror of cuda/cubic_feature_sampling/__init__.py
(CubicFeatureSamplingFunction :13-32, CubicFeatureSampling :35-42), backed by
sn_cubic_forward / sn_cubic_backward (include/sparenet_hip.h).

forward(ptcloud [B,n,3] in [-1,1], cubic_features [B,C,s,s,s], neighborhood_size=1)
    -> [B, n, (2*neighborhood_size)^3, C]: the feature vectors of the grid vertices
    around every point (zeros for vertices outside the grid).  The gradient flows to
    cubic_features only; d/d ptcloud is identically zero (floor/ceil).
"""
import torch

from sparenet_amd import _lib


class CubicFeatureSamplingFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ptcloud, cubic_features, neighborhood_size=1):
        ptcloud = ptcloud.contiguous().float()
        cubic_features = cubic_features.contiguous().float()
        b, n, _ = ptcloud.shape
        c, scale = cubic_features.size(1), cubic_features.size(2)
        ns = int(neighborhood_size)
        nv = (2 * ns) ** 3
        dev = ptcloud.device
        out = torch.empty(b, n, nv, c, device=dev)
        idx = torch.empty(b, n, nv, dtype=torch.int32, device=dev)
        with torch.cuda.device_of(ptcloud):
            code = _lib.lib().sn_cubic_forward(
                _lib.fptr(ptcloud, "ptcloud"), _lib.fptr(cubic_features, "cubic_features"),
                b, n, c, scale, ns, _lib.fptr(out, "point_features"),
                _lib.iptr(idx, "grid_pt_indexes"), _lib.stream_of(ptcloud))
        _lib.check(code, "sn_cubic_forward")
        ctx.dims = (b, n, c, scale, ns)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, grad_point_features):
        (idx,) = ctx.saved_tensors
        b, n, c, scale, ns = ctx.dims
        grad_point_features = grad_point_features.contiguous().float()
        dev = grad_point_features.device
        grad_cubic = torch.empty(b, c, scale, scale, scale, device=dev)
        with torch.cuda.device_of(grad_point_features):
            code = _lib.lib().sn_cubic_backward(
                _lib.fptr(grad_point_features, "grad_point_features"),
                _lib.iptr(idx, "grid_pt_indexes"), b, n, c, scale, ns,
                _lib.fptr(grad_cubic, "grad_cubic_features"), _lib.stream_of(grad_point_features))
        _lib.check(code, "sn_cubic_backward")
        grad_ptcloud = torch.zeros(b, n, 3, device=dev)
        return grad_ptcloud, grad_cubic, None


class CubicFeatureSampling(torch.nn.Module):
    def __init__(self):
        super(CubicFeatureSampling, self).__init__()

    def forward(self, ptcloud, cubic_features, neighborhood_size=1):
        h_scale = cubic_features.size(2) / 2
        ptcloud = ptcloud * h_scale + h_scale
        return CubicFeatureSamplingFunction.apply(ptcloud, cubic_features, neighborhood_size)
