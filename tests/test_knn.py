"""EdgeConv k-NN graph and edge features (models/sparenet_generator.py:852-906).

Golden vectors come from the reference's own functions run on the CPU (tests/golden/gen_knn.py).  The
neighbour ORDER of a row is only defined up to fp32 rounding of nearly equal distances (the reference's
two branches -- KNN_CUDA and the matmul fallback -- disagree there themselves), so indices are compared
as sets after checking that the k-th and (k+1)-th distances are separated."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _rows_match(idx_a, idx_b, x, k):
    """Same neighbour set in every row whose k-th / (k+1)-th distances are separated by more than the
    fp32 rounding of the ranking expression |x_j|^2 - 2 x_i.x_j (terms of size |x|^2, they cancel)."""
    x64 = x.astype(np.float64)
    bad = 0
    for b in range(x.shape[0]):
        xx = (x64[b] ** 2).sum(0)
        d = xx[:, None] + xx[None, :] - 2.0 * x64[b].T @ x64[b]
        srt = np.sort(d, axis=1)
        clear = (srt[:, k] - srt[:, k - 1]) > 2e-5 * (xx.max() * 3.0)
        for i in np.nonzero(clear)[0]:
            bad += set(idx_a[b, i]) != set(idx_b[b, i])
    return bad


def test_oracle_matches_reference_golden(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "knn_*.npz")))
    assert files
    for f in files:
        z = np.load(f)
        k = int(z["k"])
        idx = oracle.knn(z["x"], k)
        assert _rows_match(idx, z["idx"], z["x"], k) == 0, f
        assert (idx[:, :, 0] == np.arange(idx.shape[1])[None]).all()       # the point itself first
        np.testing.assert_array_equal(oracle.graph_feature(z["x"], z["idx"]), z["feature"])


@pytest.mark.gpu
def test_hip_knn_and_graph_feature(golden_dir, dev):
    from sparenet_amd.cuda.knn import get_graph_feature, knn

    for f in sorted(glob.glob(os.path.join(golden_dir, "knn_*.npz"))):
        z = np.load(f)
        k = int(z["k"])
        x = torch.from_numpy(z["x"]).to(dev)
        idx = knn(x, k)
        assert idx.dtype == torch.int64 and tuple(idx.shape) == z["idx"].shape
        assert _rows_match(idx.cpu().numpy(), z["idx"], z["x"], k) == 0, f
        feat = get_graph_feature(x, k=k, idx=torch.from_numpy(z["idx"]).to(dev))
        np.testing.assert_array_equal(feat.cpu().numpy(), z["feature"])


@pytest.mark.gpu
def test_hip_knn_sparenet_sizes_and_autograd(dev):
    """The generator's EdgeConv sizes (3000 points, C = 3 and 256, k = 8) against the oracle, and the
    gradient of the edge features against torch's own gather formulation."""
    from sparenet_amd.cuda.knn import get_graph_feature, knn, knn_fused, knn_unfused

    g = torch.Generator().manual_seed(5)
    for c in (3, 256):
        x = torch.rand(2, c, 3000, generator=g)
        ref = oracle.knn(x.numpy(), 8)
        for fn in (knn_fused, knn_unfused):   # the fused MFMA kernel and the GEMM + ranking pair
            idx = fn(x.to(dev), 8)
            assert _rows_match(idx.cpu().numpy(), ref, x.numpy(), 8) == 0, fn.__name__
            assert (idx[:, :, 0].cpu() == torch.arange(3000)[None]).all()
    x = torch.rand(2, 5, 300, generator=g).to(dev).requires_grad_(True)
    w = torch.rand(2, 10, 300, 4, generator=g).to(dev)
    idx = knn(x.detach(), 4)
    (get_graph_feature(x, k=4, idx=idx) * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    nb = torch.gather(x2.unsqueeze(2).expand(-1, -1, 300, -1), 3, idx.unsqueeze(1).expand(-1, 5, -1, -1))
    ref = torch.cat([nb - x2.unsqueeze(3), x2.unsqueeze(3).expand(-1, -1, -1, 4)], dim=1)
    (ref * w).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("b,c,n,k", [(1, 3, 130, 4), (2, 17, 257, 8), (3, 64, 1000, 20), (2, 5, 64, 16),
                                     (1, 512, 515, 8), (2, 33, 129, 1), (1, 7, 24, 20)])
def test_hip_fused_knn_shapes(b, c, n, k, dev):
    """Ragged sizes of the fused kernel: n not a multiple of the 128-point tile or of 4 (scalar loads),
    channel counts that do not fill a 16-channel stage, k at both list sizes, k close to n."""
    from sparenet_amd.cuda.knn import knn_fused

    g = torch.Generator().manual_seed(b * 1000 + c * 10 + k)
    x = torch.randn(b, c, n, generator=g)
    idx = knn_fused(x.to(dev), k).cpu().numpy()
    assert idx.shape == (b, n, k) and idx.min() >= 0 and idx.max() < n
    assert (idx[:, :, 0] == np.arange(n)[None]).all()
    for row in idx.reshape(-1, k):
        assert len(set(row.tolist())) == k
    assert _rows_match(idx, oracle.knn(x.numpy(), k), x.numpy(), k) == 0


@pytest.mark.gpu
def test_hip_fused_knn_ties(dev):
    """Duplicate points: equal scores resolve to the lower index, the point itself stays first."""
    from sparenet_amd.cuda.knn import knn_fused as knn

    base = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [5.0, 5.0, 5.0]]).t()   # [3, 4]
    x = base.repeat(1, 40).unsqueeze(0).contiguous()                       # [1, 3, 160]: every point 40 times
    idx = knn(x.to(dev), 8).cpu().numpy()[0]
    for i in range(160):
        same = [j for j in range(160) if j % 4 == i % 4 and j != i]
        assert idx[i, 0] == i and list(idx[i, 1:]) == same[:7], (i, idx[i])
