"""Expansion penalty: oracle vs golden vectors from the emulated reference kernel
and vs scipy's MST (CPU); HIP vs oracle / golden (GPU).  Parity bar: assignment
exact, dist and mean bit-exact."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _golden(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "expansion_*.npz")))
    assert files
    return files


def test_oracle_matches_emulated_reference_golden(golden_dir):
    for f in _golden(golden_dir):
        z = np.load(f)
        d, a, m = oracle.expansion_forward(z["xyz"], int(z["primitive_size"]), float(z["alpha"]))
        assert np.array_equal(a, z["assignment"]), f
        assert np.array_equal(d, z["dist"]), f
        assert np.array_equal(m, z["mean_mst_sum"]), f


def test_schedule_dependent_emulations_are_recorded_not_dropped(golden_dir):
    """gen_emulated.py runs the reference kernel text under several thread schedules (simt.h).  A fixture in
    which the leaf-stripping race of the last star (expansion_penalty_cuda.cu:126-135) materialises -- the
    owner of one edge per affected patch depends on the schedule -- is kept as xfail_*.npz: the race-invariant
    part (the multiset of penalised lengths, the mean MST length) must still equal the oracle's; the
    per-endpoint ownership is reported as an expected failure with the stored reason."""
    files = sorted(glob.glob(os.path.join(golden_dir, "xfail_expansion_*.npz")))
    for f in files:
        z = np.load(f)
        d, a, m = oracle.expansion_forward(z["xyz"], int(z["primitive_size"]), float(z["alpha"]))
        assert np.array_equal(np.sort(d, 1), np.sort(z["dist"], 1)), f
        assert np.array_equal(m, z["mean_mst_sum"]), f
    diverging = [f for f in files if not np.array_equal(
        oracle.expansion_forward(np.load(f)["xyz"], int(np.load(f)["primitive_size"]), float(np.load(f)["alpha"]))[1],
        np.load(f)["assignment"])]
    if diverging:
        pytest.xfail(str(np.load(diverging[0])["reason"]))


def test_oracle_mst_weight_vs_scipy():
    from scipy.sparse.csgraph import minimum_spanning_tree
    from scipy.spatial.distance import cdist

    rng = np.random.default_rng(0)
    x = rng.random((2, 256, 3), dtype=np.float32)
    P = 64
    d, a, m = oracle.expansion_forward(x, P, 1.5)
    for b in range(2):
        tot = 0.0
        for p in range(256 // P):
            pts = x[b, p * P:(p + 1) * P].astype(np.float64)
            tot += minimum_spanning_tree(cdist(pts, pts)).sum() / (P - 1)
        np.testing.assert_allclose(m[b], tot, rtol=2e-6)
    # every penalised point references a point of its own patch, with dist = edge length
    for b in range(2):
        for j in np.nonzero(a[b] >= 0)[0]:
            assert a[b, j] // P == j // P
            np.testing.assert_allclose(d[b, j], np.linalg.norm(x[b, j] - x[b, a[b, j]]), rtol=1e-6)
    assert ((a < 0) == (d == 0)).all()


def test_oracle_backward_formula():
    rng = np.random.default_rng(1)
    x = rng.random((2, 128, 3), dtype=np.float32)
    d, a, m = oracle.expansion_forward(x, 32, 1.1)
    gd = rng.random((2, 128), dtype=np.float32)
    g = oracle.expansion_backward(x, gd, a)
    ref = np.zeros_like(x)
    for b in range(2):
        for j in range(128):
            if a[b, j] >= 0:
                ref[b, j] = (gd[b, j] * 2) * (x[b, j] - x[b, a[b, j]])
    np.testing.assert_array_equal(g, ref)


def _check_mean(m, m0_sum, n, P):
    # the host mirror divides by n/P with the same torch op as the reference module (:40);
    # torch evaluates tensor/scalar on the GPU as tensor * (1/scalar): exact for 2^k patches
    m = m.detach().cpu().numpy()
    np_ = n // P
    if np_ & (np_ - 1) == 0:
        assert np.array_equal(m, (m0_sum / np.float32(np_)).astype(np.float32))
    else:
        np.testing.assert_allclose(m, m0_sum / np.float32(np_), rtol=2e-7)


def _hip(x, P, alpha, dev):
    from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyFunction

    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    d, a, m = expansionPenaltyFunction.apply(xt, P, alpha)
    return xt, d, a, m


@pytest.mark.gpu
def test_hip_matches_golden(golden_dir, dev):
    for f in _golden(golden_dir):
        z = np.load(f)
        P = int(z["primitive_size"])
        _, d, a, m = _hip(z["xyz"], P, float(z["alpha"]), dev)
        assert np.array_equal(a.cpu().numpy(), z["assignment"]), f
        assert np.array_equal(d.detach().cpu().numpy(), z["dist"]), f
        np_ = z["xyz"].shape[1] / P
        assert np.array_equal(m.detach().cpu().numpy(), (z["mean_mst_sum"] / np.float32(np_)).astype(np.float32)), f


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,P,alpha,kind", [
    (2, 1024, 512, 1.5, "uniform"), (3, 768, 256, 1.5, "uniform"), (2, 512, 128, 1.0, "uniform"),
    (4, 256, 64, 1.5, "lattice"), (2, 128, 32, 1.5, "uniform"), (2, 64, 8, 0.5, "uniform"),
    (1, 8, 4, 1.5, "uniform"), (3, 6, 2, 1.5, "uniform"), (2, 1024, 512, 1.5, "lattice"),
    (3, 1024, 512, 1.5, "jitter"), (2, 512, 256, 1.2, "jitter"), (2, 256, 64, 1.5, "jitter"),
])
def test_hip_matches_oracle(b, n, P, alpha, kind, dev):
    rng = np.random.default_rng(n + P)
    if kind == "lattice":
        x = (rng.integers(0, 6, (b, n, 3)) / 5).astype(np.float32)
    elif kind == "jitter":
        # lattice points moved by a few ulps: squared lengths that differ in their last bits while their
        # rounded square roots tie -- the kernel decides on squared lengths and must fall back to sqrtf
        x = (rng.integers(0, 8, (b, n, 3)) / 8).astype(np.float32)
        x = (x + rng.integers(-3, 4, (b, n, 3)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
    else:
        x = rng.random((b, n, 3), dtype=np.float32)
    d0, a0, m0 = oracle.expansion_forward(x, P, alpha)
    xt, d, a, m = _hip(x, P, alpha, dev)
    assert np.array_equal(a.cpu().numpy(), a0)
    assert np.array_equal(d.detach().cpu().numpy(), d0)
    _check_mean(m, m0, n, P)
    gd = rng.random((b, n), dtype=np.float32)
    (d * torch.from_numpy(gd).to(dev)).sum().backward()
    np.testing.assert_array_equal(xt.grad.cpu().numpy(), oracle.expansion_backward(x, gd, a0))


@pytest.mark.gpu
def test_hip_full_size(dev):
    """BASELINE config 2: [32,16384,3], primitive_size 512, alpha 1.5; four clouds
    are checked bit-exactly against the oracle, all of them for invariants."""
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(32, 16384, 3, generator=g)
    _, d, a, m = _hip(x.numpy(), 512, 1.5, dev)
    d, a, m = d.detach().cpu().numpy(), a.cpu().numpy(), m.detach().cpu().numpy()
    sel = [0, 7, 16, 31]
    d0, a0, m0 = oracle.expansion_forward(x[sel].numpy(), 512, 1.5)
    assert np.array_equal(a[sel], a0) and np.array_equal(d[sel], d0)
    assert np.array_equal(m[sel], (m0 / np.float32(32.0)).astype(np.float32))
    assert ((a < 0) == (d == 0)).all()
    pen = a >= 0
    assert (a[pen] // 512 == np.nonzero(pen)[1] // 512).all()

