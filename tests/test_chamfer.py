"""Chamfer distance: oracle vs golden vectors (CPU) and HIP vs oracle (GPU).

Mirrors what the reference can test about this op (cuda/chamfer_dist/test.py:22-28
is a gradcheck only); parity bar: idx exact, dist bit-exact, grads 1e-5 rel.
"""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _golden(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "chamfer_*.npz")))
    assert files, "no chamfer golden vectors"
    return files


# ------------------------------------------------------------------ CPU side
def test_oracle_matches_reference_golden(golden_dir):
    for f in _golden(golden_dir):
        z = np.load(f)
        d1, d2, i1, i2 = oracle.chamfer_forward(z["xyz1"], z["xyz2"])
        assert np.array_equal(i1, z["idx1"]) and np.array_equal(i2, z["idx2"]), f
        assert np.array_equal(d1, z["dist1"]) and np.array_equal(d2, z["dist2"]), f
        g1, g2 = oracle.chamfer_backward(z["xyz1"], z["xyz2"], z["graddist1"], z["graddist2"],
                                         z["idx1"], z["idx2"])
        assert np.array_equal(g1, z["gradxyz1"]) and np.array_equal(g2, z["gradxyz2"]), f


def test_oracle_mt_equals_single_thread():
    rng = np.random.default_rng(3)
    x = rng.random((3, 700, 3), dtype=np.float32)
    y = rng.random((3, 333, 3), dtype=np.float32)
    a = oracle.chamfer_forward(x, y)
    b = oracle.chamfer_forward(x, y, mt=True)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)


def test_oracle_vs_live_reference_build():
    """When oracle/_ref/cd_ref.so exists (built from /root/reference), check live."""
    from oracle import ref

    if not ref.available():
        pytest.skip("oracle/_ref not built here")
    g = torch.Generator().manual_seed(77)
    x = torch.rand(2, 513, 3, generator=g)
    y = torch.rand(2, 1025, 3, generator=g)
    d1, d2, i1, i2 = ref.chamfer_forward(x, y)
    o = oracle.chamfer_forward(x.numpy(), y.numpy())
    for p, q in zip(o, (d1, d2, i1, i2)):
        assert np.array_equal(p, q.numpy())


def test_host_path_matches_reference_golden_config1(golden_dir):
    """BASELINE config 1 as written: ChamferDistance fwd / bwd on CPU tensors (the branch the reference takes for
    them, chamfer_distance.py:31-32,53-54) -- the library's own host implementation (csrc/chamfer_host.hip), against
    the goldens of the reference's own CPU build, bit for bit, through the reference's class."""
    from sparenet_amd.cuda.chamfer_distance import ChamferDistanceFunction

    for f in _golden(golden_dir):
        z = np.load(f)
        x = torch.from_numpy(z["xyz1"]).requires_grad_(True)
        y = torch.from_numpy(z["xyz2"]).requires_grad_(True)
        d1, d2 = ChamferDistanceFunction.apply(x, y)
        assert not d1.is_cuda
        assert np.array_equal(d1.detach().numpy(), z["dist1"]) and np.array_equal(d2.detach().numpy(), z["dist2"]), f
        torch.autograd.backward([d1, d2], [torch.from_numpy(z["graddist1"]), torch.from_numpy(z["graddist2"])])
        assert np.array_equal(x.grad.numpy(), z["gradxyz1"]) and np.array_equal(y.grad.numpy(), z["gradxyz2"]), f


def test_host_path_edge_cases_match_oracle_and_live_reference():
    """Ragged sizes (tails of the 8-lane walk), exact ties (lowest index), duplicated points, NaN / inf coordinates
    (the reference keeps target 0 when nothing compares below it), any thread count -- against the oracle and, where
    oracle/_ref is built, against the reference's own CPU code."""
    import ctypes

    import sparenet_amd
    from oracle import ref

    lib = sparenet_amd.lib()
    rng = np.random.default_rng(11)
    cases = []
    for (b, n, m) in ((1, 1, 1), (2, 7, 9), (3, 257, 1023), (1, 1000, 8), (2, 64, 65)):
        cases.append((rng.random((b, n, 3), dtype=np.float32), rng.random((b, m, 3), dtype=np.float32)))
    lat = (rng.integers(0, 3, (2, 300, 3)) * 0.5).astype(np.float32)          # lattice: many exact ties
    cases.append((lat, np.ascontiguousarray(lat[:, ::-1][:, :211])))
    nanx, nany = rng.random((1, 40, 3), dtype=np.float32), rng.random((1, 50, 3), dtype=np.float32)
    nany[0, 0, 1] = np.nan          # target 0 NaN: every query of cloud 1 keeps it
    nany[0, 19, 0] = np.nan         # a NaN target in the middle of a lane: skipped, its lane goes on
    nanx[0, 5, 2] = np.inf
    cases.append((nanx, nany))
    for x, y in cases:
        b, n, m = x.shape[0], x.shape[1], y.shape[1]
        want = oracle.chamfer_forward(x, y)
        for threads in (1, 3, 0):
            d1, d2 = np.empty((b, n), np.float32), np.empty((b, m), np.float32)
            i1, i2 = np.empty((b, n), np.int32), np.empty((b, m), np.int32)
            vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            assert lib.sn_chamfer_forward_host(vp(x), vp(y), b, n, m, vp(d1), vp(i1), vp(d2), vp(i2), threads) == 0
            for got, w in zip((d1, d2, i1, i2), want):
                assert np.array_equal(got, w, equal_nan=True), (x.shape, y.shape, threads)
        if ref.available():
            r = ref.chamfer_forward(torch.from_numpy(x), torch.from_numpy(y))
            for got, w in zip((d1, d2, i1, i2), r):
                assert np.array_equal(got, w.numpy(), equal_nan=True)
        gd1, gd2 = rng.standard_normal((b, n)).astype(np.float32), rng.standard_normal((b, m)).astype(np.float32)
        g1w, g2w = oracle.chamfer_backward(x, y, gd1, gd2, i1, i2)
        for threads in (1, 0):
            g1, g2 = np.full((b, n, 3), 7, np.float32), np.full((b, m, 3), 7, np.float32)   # fully overwritten
            assert lib.sn_chamfer_backward_host(vp(x), vp(y), vp(gd1), vp(gd2), vp(i1), vp(i2), b, n, m, vp(g1), vp(g2),
                                                threads) == 0
            assert np.array_equal(g1, g1w, equal_nan=True) and np.array_equal(g2, g2w, equal_nan=True)
    bad = np.full((1, 4), 9, np.int32)
    z = np.zeros((1, 4, 3), np.float32)
    zz = np.zeros((1, 4), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.sn_chamfer_backward_host(vp(z), vp(z), vp(zz), vp(zz), vp(bad), vp(bad), 1, 4, 4, vp(z.copy()), vp(z.copy()), 1) == -22
    assert b"outside" in lib.sn_last_error()


def test_host_path_is_chamfer_only_and_devices_must_agree():
    from sparenet_amd.cuda.chamfer_distance import ChamferDistance, ChamferDistanceMean

    x = torch.rand(1, 33, 3)
    assert ChamferDistanceMean()(x, x * 0.5).dim() == 0
    with pytest.raises(ValueError):
        ChamferDistance()(x, torch.rand(2, 5, 3))


def test_oracle_brute_force_argmin():
    rng = np.random.default_rng(5)
    x = rng.random((2, 97, 3)).astype(np.float32)
    y = rng.random((2, 61, 3)).astype(np.float32)
    d1, d2, i1, i2 = oracle.chamfer_forward(x, y)
    dd = ((x[:, :, None, :].astype(np.float64) - y[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(i1, dd.argmin(2))
    assert np.array_equal(i2, dd.argmin(1))
    np.testing.assert_allclose(d1, dd.min(2), rtol=1e-6)


def test_host_wrapper_validates_shapes_and_gpu_entry_points_reject_cpu_tensors():
    from sparenet_amd.cuda.chamfer_distance import ChamferDistance, cd
    from sparenet_amd import SparenetHipError

    with pytest.raises(ValueError):
        ChamferDistance()(torch.rand(1, 8, 2), torch.rand(1, 8, 3))
    x = torch.rand(1, 8, 3)
    d, i = torch.empty(1, 8), torch.empty(1, 8, dtype=torch.int)
    with pytest.raises(SparenetHipError):     # the DEVICE entry points never take host memory
        cd.forward_cuda(x, x, d, d.clone(), i, i.clone())


# ------------------------------------------------------------------ GPU side
def _run_hip(x, y, gd1, gd2, dev):
    from sparenet_amd.cuda.chamfer_distance import ChamferDistanceFunction

    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    yt = torch.from_numpy(y).to(dev).requires_grad_(True)
    d1, d2 = ChamferDistanceFunction.apply(xt, yt)
    (d1 * torch.from_numpy(gd1).to(dev)).sum().add((d2 * torch.from_numpy(gd2).to(dev)).sum()).backward()
    return d1.detach().cpu().numpy(), d2.detach().cpu().numpy(), xt.grad.cpu().numpy(), yt.grad.cpu().numpy()


def _idx_hip(x, y, dev):
    from sparenet_amd.cuda.chamfer_distance import cd

    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    b, n, m = x.shape[0], x.shape[1], y.shape[1]
    d1 = torch.empty(b, n, device=dev)
    d2 = torch.empty(b, m, device=dev)
    i1 = torch.empty(b, n, dtype=torch.int, device=dev)
    i2 = torch.empty(b, m, dtype=torch.int, device=dev)
    cd.forward_cuda(xt, yt, d1, d2, i1, i2)
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


@pytest.mark.gpu
def test_hip_matches_golden(golden_dir, dev):
    for f in _golden(golden_dir):
        z = np.load(f)
        d1, d2, i1, i2 = _idx_hip(z["xyz1"], z["xyz2"], dev)
        assert np.array_equal(i1, z["idx1"]) and np.array_equal(i2, z["idx2"]), f
        assert np.array_equal(d1, z["dist1"]) and np.array_equal(d2, z["dist2"]), f
        # backward: the gather over sorted inverse lists adds the reference CPU path's terms in its order --
        # bit-equal to the goldens from the reference's own build, and bit-reproducible
        _, _, g1, g2 = _run_hip(z["xyz1"], z["xyz2"], z["graddist1"], z["graddist2"], dev)
        assert np.array_equal(g1, z["gradxyz1"]) and np.array_equal(g2, z["gradxyz2"]), f
        _, _, h1, h2 = _run_hip(z["xyz1"], z["xyz2"], z["graddist1"], z["graddist2"], dev)
        assert np.array_equal(g1, h1) and np.array_equal(g2, h2), f


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 7, 9), (3, 1024, 1023), (2, 1025, 2049),
                                   (1, 4096, 8), (5, 300, 5000)])
def test_hip_matches_oracle_ragged(b, n, m, dev):
    rng = np.random.default_rng(b * 1000 + n + m)
    x = rng.random((b, n, 3), dtype=np.float32)
    y = rng.random((b, m, 3), dtype=np.float32)
    ref_out = oracle.chamfer_forward(x, y, mt=True)
    got = _idx_hip(x, y, dev)
    for p, q in zip(got, ref_out):
        assert np.array_equal(p, q)


@pytest.mark.gpu
def test_hip_ties_lowest_index(dev):
    rng = np.random.default_rng(11)
    x = (rng.integers(0, 4, (2, 900, 3)) / 3).astype(np.float32)
    y = (rng.integers(0, 4, (2, 1100, 3)) / 3).astype(np.float32)
    y[:, 500:] = y[:, :600]  # exact duplicates far apart (different chunks and tiles)
    ref_out = oracle.chamfer_forward(x, y)
    got = _idx_hip(x, y, dev)
    for p, q in zip(got, ref_out):
        assert np.array_equal(p, q)


def _sorted_hip(x, y, dev):
    from sparenet_amd.cuda.chamfer_distance import cd

    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    b, n, m = x.shape[0], x.shape[1], y.shape[1]
    d1 = torch.empty(b, n, device=dev); d2 = torch.empty(b, m, device=dev)
    i1 = torch.empty(b, n, dtype=torch.int, device=dev); i2 = torch.empty(b, m, dtype=torch.int, device=dev)
    cd.forward_sorted_cuda(xt, yt, d1, d2, i1, i2)
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,kind", [(1, 1, 1, "uniform"), (2, 7, 9, "uniform"), (3, 1024, 1023, "uniform"),
                                        (2, 1025, 2049, "uniform"), (1, 4096, 8, "uniform"),
                                        (5, 300, 5000, "uniform"), (2, 3000, 3000, "lattice"),
                                        (2, 2500, 1500, "clustered"), (1, 2000, 2000, "far"),
                                        (2, 500, 700, "degenerate")])
def test_hip_sorted_search_matches_oracle(b, n, m, kind, dev):
    """sn_chamfer_forward_sorted (Morton sort + box pruning + MFMA filter + exact candidates) must
    return the oracle's distances and LOWEST indices bit for bit on any geometry."""
    rng = np.random.default_rng(b * 1000 + n + m)
    if kind == "uniform":
        x = rng.random((b, n, 3), dtype=np.float32); y = rng.random((b, m, 3), dtype=np.float32)
    elif kind == "lattice":      # many exact ties in distance
        x = (rng.integers(0, 6, (b, n, 3)) / 5).astype(np.float32)
        y = (rng.integers(0, 6, (b, m, 3)) / 5).astype(np.float32)
    elif kind == "clustered":    # tight clusters far from each other
        cx = rng.random((b, 8, 3)); cy = rng.random((b, 8, 3))
        x = (cx[:, rng.integers(0, 8, n)][np.arange(b), :, :] if False else
             np.stack([cx[i][rng.integers(0, 8, n)] for i in range(b)])) + 1e-3 * rng.standard_normal((b, n, 3))
        y = np.stack([cy[i][rng.integers(0, 8, m)] for i in range(b)]) + 1e-3 * rng.standard_normal((b, m, 3))
        x, y = x.astype(np.float32), y.astype(np.float32)
    elif kind == "far":          # large offsets: the |t|^2 - 2 t.q form cancels heavily
        x = (rng.random((b, n, 3)) + 100.0).astype(np.float32)
        y = (rng.random((b, m, 3)) + 100.0).astype(np.float32)
    else:                        # all points of a cloud identical / on a line
        x = np.zeros((b, n, 3), np.float32); x[..., 0] = 0.25
        y = np.zeros((b, m, 3), np.float32); y[..., 1] = np.linspace(0, 1, m, dtype=np.float32)
    ref_out = oracle.chamfer_forward(x, y, mt=True)
    got = _sorted_hip(x, y, dev)
    for p, q, name in zip(got, ref_out, ("dist1", "dist2", "idx1", "idx2")):
        assert np.array_equal(p, q), name


@pytest.mark.gpu
def test_hip_full_size_properties(dev):
    """BASELINE config 2 size [32,16384,3]: spot-check rows against the oracle and
    check size-independent properties on everything."""
    from sparenet_amd.cuda.chamfer_distance import cd

    g = torch.Generator().manual_seed(1234)
    x = torch.rand(32, 16384, 3, generator=g)
    y = torch.rand(32, 16384, 3, generator=g)
    xt, yt = x.to(dev), y.to(dev)
    d1 = torch.empty(32, 16384, device=dev)
    d2 = torch.empty_like(d1)
    i1 = torch.empty(32, 16384, dtype=torch.int, device=dev)
    i2 = torch.empty_like(i1)
    cd.forward_cuda(xt, yt, d1, d2, i1, i2)
    # property: dist equals the distance to the reported index, recomputed
    sel = torch.gather(yt, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3))
    diff = sel - xt
    rec = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal(rec, d1)
    # property: symmetric swap gives swapped outputs
    e1 = torch.empty_like(d1); e2 = torch.empty_like(d1)
    j1 = torch.empty_like(i1); j2 = torch.empty_like(i1)
    cd.forward_cuda(yt, xt, e1, e2, j1, j2)
    assert torch.equal(e1, d2) and torch.equal(e2, d1) and torch.equal(j1, i2) and torch.equal(j2, i1)
    # two whole batch elements bit-exact against the oracle
    o = oracle.chamfer_forward(x[[0, 31]].numpy(), y[[0, 31]].numpy(), mt=True)
    assert np.array_equal(o[0], d1[[0, 31]].cpu().numpy())
    assert np.array_equal(o[2], i1[[0, 31]].cpu().numpy())
    assert np.array_equal(o[1], d2[[0, 31]].cpu().numpy())
    assert np.array_equal(o[3], i2[[0, 31]].cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,kind", [(2, 3000, 1700, "uniform"), (1, 5000, 5000, "far"), (3, 513, 2049, "lattice"),
                                        (1, 20000, 7, "uniform"), (1, 30000, 9000, "uniform"),
                                        (2, 18000, 1, "uniform"), (2, 16384, 16384, "collapsed")])
def test_hip_backward_bit_exact_incl_long_inverse_lists(b, n, m, kind, dev):
    """Backward against the oracle (= the reference CPU order), bit for bit.  "far": every query of one cloud
    shares ONE neighbour in the other (an inverse list of thousands of entries: the heap-sort path); the
    7-point cloud gives lists of ~3000 entries each; 30000 + 9000 points exceed the LDS counters of the
    one-launch list builder (the three generic kernels run); a 1-point cloud gives one list of 18000 entries (beyond
    the LDS sort of chamfer_bwd_long_kernel: the in-place fallback); "collapsed" is the early-training picture (every
    prediction within 1e-3 of one point: a few targets collect thousands of queries each)."""
    rng = np.random.default_rng(n + m)
    x = rng.random((b, n, 3), dtype=np.float32)
    y = rng.random((b, m, 3), dtype=np.float32)
    if kind == "far":
        y = y * 0.01 + 40.0
    if kind == "collapsed":
        x = (0.5 + 1e-3 * (x - 0.5)).astype(np.float32)
    if kind == "lattice":
        x = (rng.integers(0, 5, (b, n, 3)) / 4).astype(np.float32)
        y = (rng.integers(0, 5, (b, m, 3)) / 4).astype(np.float32)
    gd1 = rng.random((b, n), dtype=np.float32)
    gd2 = rng.random((b, m), dtype=np.float32)
    _, _, i1, i2 = oracle.chamfer_forward(x, y, mt=True)
    r1, r2 = oracle.chamfer_backward(x, y, gd1, gd2, i1, i2)
    _, _, g1, g2 = _run_hip(x, y, gd1, gd2, dev)
    assert np.array_equal(g1, r1) and np.array_equal(g2, r2)
