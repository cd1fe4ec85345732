"""Gridding loss -- host-side mirror of cuda/gridding_loss/__init__.py
(GriddingDistanceFunction :13-45, GriddingDistance :48-97, GriddingLoss :100-122), backed by
sn_gridding_dist_forward / sn_gridding_backward (include/sparenet_hip.h).

GriddingDistance(scale)(pred [B,n,3], gt [B,m,3]) grids both clouds over their common integer
bounding box (every vertex keeps one trilinear accumulator per corner role -> [B, nverts, 8]);
GriddingLoss(scales, alphas) is the alpha-weighted sum of L1 distances between the two grids.
"""
import torch

from sparenet_amd import _lib


def _grid_one(cloud, bounds):
    cloud = cloud.contiguous().float()
    b, n, _ = cloud.shape
    mnx, mxx, mny, mxy, mnz, mxz = bounds
    nverts = (mxx - mnx + 1) * (mxy - mny + 1) * (mxz - mnz + 1)
    dev = cloud.device
    grid = torch.empty(b, nverts, 8, device=dev)
    weights = torch.empty(b, n, 8, 3, device=dev)
    indexes = torch.empty(b, n, 8, dtype=torch.int32, device=dev)
    with torch.cuda.device_of(cloud):
        code = _lib.lib().sn_gridding_dist_forward(
            _lib.fptr(cloud, "ptcloud"), b, n, mnx, mxx, mny, mxy, mnz, mxz, _lib.fptr(grid, "grid"),
            _lib.fptr(weights, "grid_pt_weights"), _lib.iptr(indexes, "grid_pt_indexes"),
            _lib.stream_of(cloud))
    _lib.check(code, "sn_gridding_dist_forward")
    return grid, weights, indexes


def _grad_one(grad_grid, weights, indexes):
    grad_grid = grad_grid.contiguous().float()
    b, n = indexes.shape[:2]
    grad_cloud = torch.empty(b, n, 3, device=grad_grid.device)
    with torch.cuda.device_of(grad_grid):
        code = _lib.lib().sn_gridding_backward(
            _lib.fptr(grad_grid, "grad_grid"), _lib.fptr(weights, "grid_pt_weights"),
            _lib.iptr(indexes, "grid_pt_indexes"), b, n, grad_grid.size(1) * grad_grid.size(2),
            _lib.fptr(grad_cloud, "grad_ptcloud"), _lib.stream_of(grad_grid))
    _lib.check(code, "sn_gridding_backward")
    return grad_cloud


class GriddingDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, min_x, max_x, min_y, max_y, min_z, max_z, pred_cloud, gt_cloud):
        bounds = tuple(int(v) for v in (min_x, max_x, min_y, max_y, min_z, max_z))
        pred_grid, pw, pi = _grid_one(pred_cloud, bounds)
        gt_grid, gw, gi = _grid_one(gt_cloud, bounds)
        ctx.save_for_backward(pw, pi, gw, gi)
        return pred_grid, gt_grid

    @staticmethod
    def backward(ctx, grad_pred_grid, grad_gt_grid):
        pw, pi, gw, gi = ctx.saved_tensors
        return (None, None, None, None, None, None, _grad_one(grad_pred_grid, pw, pi),
                _grad_one(grad_gt_grid, gw, gi))


class GriddingDistance(torch.nn.Module):
    def __init__(self, scale=1):
        super().__init__()
        self.scale = scale

    def forward(self, pred_cloud, gt_cloud):
        """pred_cloud [B,n,3], gt_cloud [B,m,3] in [-1,1] -> (pred_grid, gt_grid) [B, nverts, 8]."""
        pred_cloud = pred_cloud * self.scale / 2
        gt_cloud = gt_cloud * self.scale / 2
        # one integer box for the whole batch and both clouds, one vertex of margin
        # (six reductions + one host read instead of the reference's twelve .min()/.max() calls)
        both = torch.cat([pred_cloud.reshape(-1, 3), gt_cloud.reshape(-1, 3)], dim=0)
        lo = (torch.floor(both.min(dim=0).values) - 1).tolist()
        hi = (torch.ceil(both.max(dim=0).values) + 1).tolist()
        bounds = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
        pred_grids, gt_grids = [], []
        for pc, gc in zip(pred_cloud.split(1, dim=0), gt_cloud.split(1, dim=0)):
            pc = pc[torch.sum(pc, dim=2).ne(0)].unsqueeze(dim=0)   # zero rows are padding
            gc = gc[torch.sum(gc, dim=2).ne(0)].unsqueeze(dim=0)
            pg, gg = GriddingDistanceFunction.apply(*bounds, pc, gc)
            pred_grids.append(pg)
            gt_grids.append(gg)
        return torch.cat(pred_grids, dim=0).contiguous(), torch.cat(gt_grids, dim=0).contiguous()


class GriddingLoss(torch.nn.Module):
    def __init__(self, scales=[], alphas=[]):
        super().__init__()
        self.scales = scales
        self.alphas = alphas
        self.gridding_dists = [GriddingDistance(scale=s) for s in scales]
        self.l1_loss = torch.nn.L1Loss()

    def forward(self, pred_cloud, gt_cloud):
        total = None
        for alpha, gdist in zip(self.alphas, self.gridding_dists):
            pred_grid, gt_grid = gdist(pred_cloud, gt_cloud)
            term = alpha * self.l1_loss(pred_grid, gt_grid)
            total = term if total is None else total + term
        return total
