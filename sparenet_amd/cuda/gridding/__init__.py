"""Gridding / GriddingReverse -- host-side mirror of cuda/gridding/__init__.py
(GriddingFunction :13-32, Gridding :35-49, GriddingReverseFunction :52-65,
GriddingReverse :68-75), backed by sn_gridding_* (include/sparenet_hip.h).

Gridding(scale)(ptcloud [B,n,3] in [-1,1)) -> trilinear occupancy grid [B, scale^3];
all-zero (padding) points are dropped per sample, as in the reference.
GriddingReverse(scale)(grid [B,s,s,s]) -> point per grid cell [B, s^3, 3].
"""
import torch

from sparenet_amd import _lib


class GriddingFunction(torch.autograd.Function):
    """scale here is the HALF scale the reference passes (bounds [-scale, scale-1]).  With
    skip_padding the rows whose coordinates sum to zero contribute nothing (zero gradient): the
    module's padding rule, applied in the kernel."""

    @staticmethod
    def forward(ctx, scale, ptcloud, skip_padding=False):
        ptcloud = ptcloud.contiguous().float()
        b, n, _ = ptcloud.shape
        full = 2 * int(scale)
        dev = ptcloud.device
        grid = torch.empty(b, full ** 3, device=dev)
        weights = torch.empty(b, n, 8, 3, device=dev)
        indexes = torch.empty(b, n, 8, dtype=torch.int32, device=dev)
        with torch.cuda.device_of(ptcloud):
            fn = _lib.lib().sn_gridding_forward_padded if skip_padding else _lib.lib().sn_gridding_forward
            code = fn(
                _lib.fptr(ptcloud, "ptcloud"), b, n, full, _lib.fptr(grid, "grid"),
                _lib.fptr(weights, "grid_pt_weights"), _lib.iptr(indexes, "grid_pt_indexes"),
                _lib.stream_of(ptcloud))
        _lib.check(code, "sn_gridding_forward")
        ctx.save_for_backward(weights, indexes)
        return grid

    @staticmethod
    def backward(ctx, grad_grid):
        weights, indexes = ctx.saved_tensors
        grad_grid = grad_grid.contiguous().float()
        b, n = indexes.shape[:2]
        grad_ptcloud = torch.empty(b, n, 3, device=grad_grid.device)
        with torch.cuda.device_of(grad_grid):
            code = _lib.lib().sn_gridding_backward(
                _lib.fptr(grad_grid, "grad_grid"), _lib.fptr(weights, "grid_pt_weights"),
                _lib.iptr(indexes, "grid_pt_indexes"), b, n, grad_grid.size(1),
                _lib.fptr(grad_ptcloud, "grad_ptcloud"), _lib.stream_of(grad_grid))
        _lib.check(code, "sn_gridding_backward")
        return None, grad_ptcloud, None


class Gridding(torch.nn.Module):
    """ptcloud [B,n,3] in [-1,1) -> occupancy grid [B, scale^3].  Rows whose coordinates sum
    to zero are padding and are ignored (cuda/gridding/__init__.py:41-47)."""

    def __init__(self, scale=1):
        super().__init__()
        self.scale = scale // 2          # half extent: vertices span [-scale/2, scale/2 - 1]

    def forward(self, ptcloud):
        # the reference grids sample by sample after dropping the zero (padding) rows on the host;
        # the kernel applies the same rule per point, so the whole batch is one launch
        return GriddingFunction.apply(self.scale, ptcloud * self.scale, True)


class GriddingReverseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, grid):
        grid = grid.contiguous().float()
        b = grid.size(0)
        ptcloud = torch.empty(b, scale ** 3, 3, device=grid.device)
        with torch.cuda.device_of(grid):
            code = _lib.lib().sn_gridding_reverse_forward(
                _lib.fptr(grid, "grid"), b, int(scale), _lib.fptr(ptcloud, "ptcloud"),
                _lib.stream_of(grid))
        _lib.check(code, "sn_gridding_reverse_forward")
        ctx.scale = int(scale)
        ctx.save_for_backward(grid, ptcloud)
        return ptcloud

    @staticmethod
    def backward(ctx, grad_ptcloud):
        grid, ptcloud = ctx.saved_tensors
        scale = ctx.scale
        grad_ptcloud = grad_ptcloud.contiguous().float()
        b = grid.size(0)
        grad_grid = torch.empty(b, scale ** 3, device=grid.device)
        with torch.cuda.device_of(grid):
            code = _lib.lib().sn_gridding_reverse_backward(
                _lib.fptr(grad_ptcloud, "grad_ptcloud"), _lib.fptr(grid, "grid"),
                _lib.fptr(ptcloud, "ptcloud"), b, scale, _lib.fptr(grad_grid, "grad_grid"),
                _lib.stream_of(grid))
        _lib.check(code, "sn_gridding_reverse_backward")
        return None, grad_grid.view(-1, scale, scale, scale)


class GriddingReverse(torch.nn.Module):
    """grid [B,s,s,s] -> one point per interior vertex [B, s^3, 3], rescaled to [-1,1)."""

    def __init__(self, scale=1):
        super().__init__()
        self.scale = scale

    def forward(self, grid):
        vertex_space = GriddingReverseFunction.apply(self.scale, grid)
        return vertex_space / self.scale * 2
