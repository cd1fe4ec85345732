"""Gridding / GriddingReverse / CubicFeatureSampling.

CPU: oracle vs golden vectors from the reference kernel text run by the SIMT emulator.
GPU: HIP vs oracle and golden: indices/weights/single-writer outputs exact, atomic
sums within 1e-5; gradients also vs finite differences in the spirit of the reference's
gradcheck tests (cuda/gridding/test.py:23-41, cuda/cubic_feature_sampling/test.py:23-56).
"""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _golden(golden_dir, pat):
    files = sorted(glob.glob(os.path.join(golden_dir, pat)))
    assert files
    return files


def test_oracle_matches_emulated_reference_golden(golden_dir):
    for f in _golden(golden_dir, "gridding_*.npz"):
        z = np.load(f)
        scale = int(z["scale"])
        g, w, ix = oracle.gridding_forward(z["ptcloud"], scale)
        assert np.array_equal(w, z["weights"]) and np.array_equal(ix, z["indexes"]), f
        np.testing.assert_allclose(g, z["grid"], rtol=1e-5, atol=1e-6)
        assert np.array_equal(oracle.gridding_backward(z["grad_grid"], w, ix), z["grad_ptcloud"]), f
        rp = oracle.gridding_reverse_forward(z["rev_grid"], scale)
        assert np.array_equal(rp, z["rev_ptcloud"]), f
        rg = oracle.gridding_reverse_backward(z["rev_grad_ptcloud"], z["rev_grid"], rp, scale)
        np.testing.assert_allclose(rg.reshape(rg.shape[0], -1), z["rev_grad_grid"], rtol=1e-4, atol=1e-5)
    for f in _golden(golden_dir, "cubic_*.npz"):
        z = np.load(f)
        ns = int(z["neighborhood_size"])
        out, ix = oracle.cubic_forward(z["ptcloud"], z["feat"], ns)
        assert np.array_equal(out, z["out"]) and np.array_equal(ix, z["indexes"]), f
        c, scale = z["feat"].shape[1], z["feat"].shape[2]
        np.testing.assert_allclose(oracle.cubic_backward(z["grad_out"], ix, c, scale, ns),
                                   z["grad_feat"], rtol=1e-5, atol=1e-6)


def test_oracle_gridding_partition_of_unity():
    rng = np.random.default_rng(0)
    pt = ((rng.random((2, 100, 3)) * 1.4 - 0.7) * 4).astype(np.float32)  # stay inside [-s, s-1)
    g, w, ix = oracle.gridding_forward(pt, 8)
    np.testing.assert_allclose(g.sum(1), 100, rtol=1e-5)
    np.testing.assert_allclose(w.prod(-1).sum(-1), 1, rtol=1e-5)


# ------------------------------------------------------------------ GPU side
@pytest.mark.gpu
def test_hip_matches_golden_and_oracle(golden_dir, dev):
    from sparenet_amd.cuda.gridding import GriddingFunction, GriddingReverseFunction
    from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSamplingFunction

    for f in _golden(golden_dir, "gridding_*.npz"):
        z = np.load(f)
        scale = int(z["scale"])
        pt = torch.from_numpy(z["ptcloud"]).to(dev).requires_grad_(True)
        grid = GriddingFunction.apply(scale // 2, pt)
        np.testing.assert_allclose(grid.detach().cpu().numpy(), z["grid"], rtol=1e-5, atol=1e-6)
        (grid * torch.from_numpy(z["grad_grid"]).to(dev)).sum().backward()
        assert np.array_equal(pt.grad.cpu().numpy(), z["grad_ptcloud"]), f
        rg = torch.from_numpy(z["rev_grid"]).to(dev).view(-1, scale, scale, scale).requires_grad_(True)
        rp = GriddingReverseFunction.apply(scale, rg)
        assert np.array_equal(rp.detach().cpu().numpy(), z["rev_ptcloud"]), f
        (rp * torch.from_numpy(z["rev_grad_ptcloud"]).to(dev)).sum().backward()
        np.testing.assert_allclose(rg.grad.cpu().numpy().reshape(rg.shape[0], -1), z["rev_grad_grid"],
                                   rtol=1e-4, atol=1e-5)
    for f in _golden(golden_dir, "cubic_*.npz"):
        z = np.load(f)
        ns = int(z["neighborhood_size"])
        ft = torch.from_numpy(z["feat"]).to(dev).requires_grad_(True)
        out = CubicFeatureSamplingFunction.apply(torch.from_numpy(z["ptcloud"]).to(dev), ft, ns)
        assert np.array_equal(out.detach().cpu().numpy(), z["out"]), f
        (out * torch.from_numpy(z["grad_out"]).to(dev)).sum().backward()
        np.testing.assert_allclose(ft.grad.cpu().numpy(), z["grad_feat"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_modules_grnet_shapes(dev):
    """Module-level API at GRNet's sizes (models/grnet_generator.py:58-99): scale 64 grid,
    2048-point clouds with zero padding, features at 32/16/8 voxels."""
    from sparenet_amd.cuda.gridding import Gridding, GriddingReverse
    from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSampling

    g = torch.Generator().manual_seed(3)
    pc = (torch.rand(4, 2048, 3, generator=g) * 1.8 - 0.9) * 0.95
    pc[:, 2000:] = 0  # padded rows are dropped
    grid = Gridding(scale=64)(pc.to(dev))
    assert grid.shape == (4, 64 ** 3)
    np.testing.assert_allclose(grid.sum(1).cpu().numpy(), 2000, rtol=1e-4)
    og, _, _ = oracle.gridding_forward((pc[:, :2000] * 32).numpy(), 64)
    np.testing.assert_allclose(grid.cpu().numpy(), og, rtol=1e-5, atol=1e-6)
    pts = GriddingReverse(scale=64)(grid.view(4, 64, 64, 64))
    assert pts.shape == (4, 64 ** 3, 3)
    orp = oracle.gridding_reverse_forward(grid.cpu().numpy(), 64)
    np.testing.assert_allclose(pts.cpu().numpy(), orp / 64 * 2, rtol=1e-6, atol=1e-7)
    for c, s in ((32, 32), (64, 16), (128, 8)):
        feat = torch.rand(4, c, s, s, s, generator=g)
        out = CubicFeatureSampling()(pc.to(dev), feat.to(dev))
        assert out.shape == (4, 2048, 8, c)
        oo, _ = oracle.cubic_forward((pc * (s / 2) + s / 2).numpy(), feat.numpy(), 1)
        assert np.array_equal(out.cpu().numpy(), oo)


@pytest.mark.gpu
def test_hip_gridding_gradients_finite_difference(dev):
    from sparenet_amd.cuda.gridding import GriddingFunction, GriddingReverseFunction

    g = torch.Generator().manual_seed(8)
    pt = ((torch.rand(1, 32, 3, generator=g) * 1.6 - 0.8) * 4 + 0.013).to(dev)
    wgt = torch.rand(1, 512, generator=g).to(dev)
    p = pt.clone().requires_grad_(True)
    (GriddingFunction.apply(4, p) * wgt).sum().backward()
    eps = 1e-2
    for (j, a) in ((0, 0), (5, 1), (17, 2)):
        pp, pm = pt.clone(), pt.clone()
        pp[0, j, a] += eps
        pm[0, j, a] -= eps
        fd = ((GriddingFunction.apply(4, pp) * wgt).sum() - (GriddingFunction.apply(4, pm) * wgt).sum()) / (2 * eps)
        assert abs(float(fd) - float(p.grad[0, j, a])) < 2e-2 * max(1.0, abs(float(fd)))
    grid = torch.rand(1, 4, 4, 4, generator=g).to(dev) + 0.1
    w3 = torch.rand(1, 64, 3, generator=g).to(dev)
    gr = grid.clone().requires_grad_(True)
    (GriddingReverseFunction.apply(4, gr) * w3).sum().backward()
    for idx in ((0, 1, 2, 3), (0, 3, 3, 3), (0, 0, 0, 0)):
        gp, gm = grid.clone(), grid.clone()
        gp[idx] += 1e-2
        gm[idx] -= 1e-2
        fd = ((GriddingReverseFunction.apply(4, gp) * w3).sum() - (GriddingReverseFunction.apply(4, gm) * w3).sum()) / 2e-2
        assert abs(float(fd) - float(gr.grad[idx])) < 3e-2 * max(1.0, abs(float(fd)))


# ------------------------------------------------------------------ gridding distance / loss
def test_oracle_gridding_dist_matches_emulated_reference_golden(golden_dir):
    """oracle/gridding.c vs the reference gridding_distance.cu kernels run under the SIMT emulator
    (tests/golden/gen_emulated.py gridding_dist)."""
    import glob

    files = sorted(glob.glob(os.path.join(golden_dir, "griddist_*.npz")))
    assert files
    for f in files:
        z = np.load(f)
        g, w, ix = oracle.gridding_dist_forward(z["ptcloud"], z["bounds"])
        assert np.array_equal(w, z["weights"]) and np.array_equal(ix, z["indexes"]), f
        np.testing.assert_allclose(g, z["grid"], rtol=1e-5, atol=1e-6, err_msg=f)
        gp = oracle.gridding_backward(z["grad_grid"], z["weights"], z["indexes"])
        np.testing.assert_allclose(gp, z["grad_ptcloud"], rtol=1e-5, atol=1e-6, err_msg=f)


@pytest.mark.gpu
def test_hip_gridding_dist_matches_golden(golden_dir, dev):
    import glob

    from sparenet_amd.cuda.gridding_loss import _grad_one, _grid_one

    for f in sorted(glob.glob(os.path.join(golden_dir, "griddist_*.npz"))):
        z = np.load(f)
        grid, w, ix = _grid_one(torch.from_numpy(z["ptcloud"]).to(dev), tuple(int(v) for v in z["bounds"]))
        assert np.array_equal(w.cpu().numpy(), z["weights"]) and np.array_equal(ix.cpu().numpy(), z["indexes"]), f
        np.testing.assert_allclose(grid.cpu().numpy(), z["grid"], rtol=1e-5, atol=1e-6, err_msg=f)
        gg = torch.from_numpy(z["grad_grid"]).to(dev).view(grid.shape)
        gp = _grad_one(gg, w, ix)
        np.testing.assert_allclose(gp.cpu().numpy(), z["grad_ptcloud"], rtol=1e-5, atol=1e-6, err_msg=f)


@pytest.mark.gpu
def test_hip_gridding_loss_module(dev):
    """GriddingLoss mirrors cuda/gridding_loss/__init__.py:100-122: zero for identical clouds, grids
    carry one unit of weight per point, gradients flow to the prediction only through its grid."""
    from sparenet_amd.cuda.gridding_loss import GriddingDistance, GriddingLoss

    g = torch.Generator().manual_seed(4)
    gt = (torch.rand(2, 400, 3, generator=g) * 1.8 - 0.9).to(dev)
    pred = (gt + 0.05 * torch.randn(2, 400, 3, generator=g).to(dev)).requires_grad_(True)
    pg, gg = GriddingDistance(scale=16)(pred, gt)
    assert pg.shape == gg.shape and pg.shape[2] == 8
    np.testing.assert_allclose(pg.sum(dim=(1, 2)).detach().cpu().numpy(), [400.0, 400.0], rtol=1e-4)
    loss_fn = GriddingLoss(scales=[16, 8], alphas=[0.1, 0.01])
    assert float(loss_fn(gt, gt)) < 1e-7          # two griddings of one cloud differ only by fp32 atomic order
    loss = loss_fn(pred, gt)
    loss.backward()
    assert float(loss) > 0 and torch.isfinite(pred.grad).all() and float(pred.grad.abs().sum()) > 0
