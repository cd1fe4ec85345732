"""GRNet-style Chamfer loss -- host-side mirror of cuda/chamfer_dist/__init__.py
(ChamferFunction :6-18, ChamferDistance :21-35, ChamferDistanceSeperate :38-52).

The reference builds a second copy of the Chamfer kernels for this module (`chamfer`
extension, cuda/chamfer_dist/chamfer.cu); here it is the same HIP path as
sparenet_amd.cuda.chamfer_distance (sn_chamfer_forward[_sorted] / sn_chamfer_backward).
"""
import torch

from sparenet_amd.cuda.chamfer_distance.chamfer_distance import ChamferDistanceFunction

ChamferFunction = ChamferDistanceFunction


def _drop_padding(xyz1, xyz2, ignore_zeros):
    """With batch size 1, rows whose coordinates sum to zero are padding (reference :27-31)."""
    if xyz1.size(0) == 1 and ignore_zeros:
        xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(dim=0)
        xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(dim=0)
    return xyz1, xyz2


class ChamferDistance(torch.nn.Module):
    """mean_j dist1 + mean_k dist2 (squared distances, both directions)."""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        dist1, dist2 = ChamferFunction.apply(*_drop_padding(xyz1, xyz2, self.ignore_zeros))
        return torch.mean(dist1) + torch.mean(dist2)


class ChamferDistanceSeperate(torch.nn.Module):
    """The two directed terms separately (the reference's spelling is kept)."""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        dist1, dist2 = ChamferFunction.apply(*_drop_padding(xyz1, xyz2, self.ignore_zeros))
        return torch.mean(dist1), torch.mean(dist2)
