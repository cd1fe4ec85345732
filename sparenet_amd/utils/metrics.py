"""Validation metrics from one pass of the loss kernels.

The reference's `Metrics` (utils/misc.py:119-211) evaluates, per validation sample,
  F-Score@th   through open3d on the CPU: nearest-neighbour distances both ways, then
               precision = #{d(pred -> gt) < th} / |pred|, recall = #{d(gt -> pred) < th} / |gt|,
               F = 2 P R / (P + R)                                     (:178-190)
  ChamferDistance x 1000                                              (:198-201)
  EMD x 100 = mean sqrt(dist) of emdModule(eps 0.005, 50 iterations)  (:203-209)
The nearest-neighbour distances of the F-score are exactly the square roots of the Chamfer
distances the second metric computes anyway, so one Chamfer forward serves both and nothing
leaves the GPU.  d < th is evaluated as dist < th^2 on the fp32 squared distances (open3d works in
double on the same fp32 coordinates: the two can only disagree for |d - th| ~ 1e-9).
"""
import torch

from sparenet_amd.cuda.chamfer_distance.chamfer_distance import ChamferDistanceFunction
from sparenet_amd.cuda.emd.emd_module import emdModule


def f_score_from_chamfer(dist1, dist2, th=0.01):
    """dist1 [B,N] = squared NN distance pred -> gt, dist2 [B,M] = gt -> pred; returns [B]."""
    th2 = float(th) * float(th)
    precision = (dist1 < th2).double().mean(dim=1)
    recall = (dist2 < th2).double().mean(dim=1)
    denom = precision + recall
    return torch.where(denom > 0, 2 * precision * recall / denom.clamp_min(1e-300),
                       torch.zeros_like(denom))


def fused_validation_metrics(pred, gt, th=0.01, emd_eps=0.005, emd_iters=50, with_emd=True):
    """pred [B,N,3], gt [B,M,3] on the GPU -> dict of per-sample tensors [B]:
    'F-Score', 'ChamferDistance' (x1000, mean dist1 + mean dist2, utils/misc.py:198-201 with
    ChamferDistanceMean) and 'EMD' (x100; needs N == M, a multiple of 1024)."""
    dist1, dist2 = ChamferDistanceFunction.apply(pred, gt)
    out = {"F-Score": f_score_from_chamfer(dist1, dist2, th),
           "ChamferDistance": (dist1.mean(dim=1) + dist2.mean(dim=1)) * 1000}
    if with_emd:
        dist, _ = emdModule()(pred, gt, emd_eps, emd_iters)
        out["EMD"] = torch.sqrt(dist).mean(dim=1) * 100
    return out
