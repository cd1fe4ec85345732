"""GRNet-style chamfer_dist module and the fused validation metrics.

The F-score oracle here is the reference's definition (utils/misc.py:178-190) evaluated with an exact
k-d tree in double precision -- what open3d's compute_point_cloud_distance returns."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree


def _f_score_ref(pred, gt, th):
    d1 = cKDTree(gt.astype(np.float64)).query(pred.astype(np.float64))[0]
    d2 = cKDTree(pred.astype(np.float64)).query(gt.astype(np.float64))[0]
    recall = float((d2 < th).sum()) / len(d2)
    precision = float((d1 < th).sum()) / len(d1)
    return 2 * recall * precision / (recall + precision) if recall + precision else 0.0


def test_f_score_formula_cpu():
    from sparenet_amd.utils.metrics import f_score_from_chamfer

    d1 = torch.tensor([[1e-5, 2e-4, 0.5, 9.9e-5]])    # squared distances
    d2 = torch.tensor([[0.0, 1.0]])
    f = f_score_from_chamfer(d1, d2, th=0.01)          # th^2 = 1e-4: precision 2/4, recall 1/2
    assert abs(float(f) - 0.5) < 1e-12
    assert float(f_score_from_chamfer(torch.ones(1, 3), torch.ones(1, 3))) == 0.0


@pytest.mark.gpu
def test_fused_metrics_match_definitions(dev):
    from sparenet_amd.utils.metrics import fused_validation_metrics

    rng = np.random.default_rng(3)
    gt = rng.random((2, 2048, 3), dtype=np.float32) - 0.5
    pred = (gt[:, rng.permutation(2048)] + 0.004 * rng.standard_normal((2, 2048, 3))).astype(np.float32)
    m = fused_validation_metrics(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), th=0.01)
    import oracle
    o1, o2, _, _ = oracle.chamfer_forward(pred, gt)          # bit-equal to the HIP distances (tests/test_chamfer.py)
    th2 = 0.01 * 0.01
    for b in range(2):
        # exact: the threshold counts are integer work on distances that are bit-equal to the oracle's
        p_cnt, r_cnt = int((o1[b] < th2).sum()), int((o2[b] < th2).sum())
        prec, rec = p_cnt / o1.shape[1], r_cnt / o2.shape[1]
        want = 2 * prec * rec / (prec + rec) if prec + rec else 0.0
        assert abs(float(m["F-Score"][b]) - want) < 1e-12, (p_cnt, r_cnt)
        # and an independent definition (exact k-d tree in double): counts may differ only for points whose
        # distance is within rounding of the threshold
        assert abs(float(m["F-Score"][b]) - _f_score_ref(pred[b], gt[b], 0.01)) < 1e-3
        d1 = cKDTree(gt[b].astype(np.float64)).query(pred[b].astype(np.float64))[0] ** 2
        d2 = cKDTree(pred[b].astype(np.float64)).query(gt[b].astype(np.float64))[0] ** 2
        np.testing.assert_allclose(float(m["ChamferDistance"][b]), (d1.mean() + d2.mean()) * 1000, rtol=1e-5)
        assert 0.0 < float(m["EMD"][b]) < 100.0


@pytest.mark.gpu
def test_chamfer_dist_module_matches_chamfer_distance(dev):
    from sparenet_amd.cuda.chamfer_dist import ChamferDistance, ChamferDistanceSeperate
    from sparenet_amd.cuda.chamfer_distance import ChamferDistanceMean

    g = torch.Generator().manual_seed(9)
    x = torch.rand(3, 500, 3, generator=g).to(dev).requires_grad_(True)
    y = torch.rand(3, 700, 3, generator=g).to(dev)
    a = ChamferDistance()(x, y)
    b = ChamferDistanceMean()(x, y)
    assert torch.equal(a, b)
    s1, s2 = ChamferDistanceSeperate()(x, y)
    assert torch.equal(s1 + s2, a)
    a.backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    # ignore_zeros only acts on batch size 1: padded rows (all-zero points) are dropped
    x1 = torch.rand(1, 300, 3, generator=g)
    x1[0, 100:150] = 0
    y1 = torch.rand(1, 200, 3, generator=g)
    full = ChamferDistance(ignore_zeros=True)(x1.to(dev), y1.to(dev))
    keep = x1[0][x1[0].sum(1) != 0].unsqueeze(0)
    assert torch.equal(full, ChamferDistance()(keep.to(dev), y1.to(dev)))
