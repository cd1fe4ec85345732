"""EMD (auction): oracle vs golden vectors from the emulated reference kernels
(CPU) and HIP vs oracle / golden (GPU).  Parity bar: assignment exact, dist
bit-exact, gradient bit-exact (single writer per element)."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _golden(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "emd_*.npz")))
    assert files
    return files


def _clouds(b, n, seed, kind="uniform"):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(b, n, 3, generator=g)
    if kind == "near":
        perm = torch.randperm(n, generator=g)
        y = (x + 0.01 * torch.randn(b, n, 3, generator=g))[:, perm].clamp(0, 1)
    elif kind == "lattice":
        x = torch.randint(0, 8, (b, n, 3), generator=g).float() / 7
        y = torch.randint(0, 8, (b, n, 3), generator=g).float() / 7
    elif kind == "clustered":
        c = torch.rand(b, 6, 3, generator=g)
        pick = lambda: torch.gather(c, 1, torch.randint(0, 6, (b, n, 1), generator=g).expand(-1, -1, 3))
        x = (pick() + 0.003 * torch.randn(b, n, 3, generator=g)).clamp(0, 1)
        y = (pick() + 0.003 * torch.randn(b, n, 3, generator=g)).clamp(0, 1)
    elif kind == "far":
        x = x + 50.0
        y = torch.rand(b, n, 3, generator=g) + 50.0
    else:
        y = torch.rand(b, n, 3, generator=g)
    return x.numpy(), y.numpy()


# ------------------------------------------------------------------ CPU side
def test_oracle_matches_emulated_reference_golden(golden_dir):
    for f in _golden(golden_dir):
        z = np.load(f)
        if int(z["iters"]) > 20:
            continue  # the it50 case is covered on the GPU side (keeps CPU suite short)
        d, a, aux = oracle.emd_forward(z["xyz1"], z["xyz2"], float(z["eps"]), int(z["iters"]),
                                       return_aux=True)
        assert np.array_equal(a, z["assignment"]), f
        assert np.array_equal(d, z["dist"]), f
        assert np.array_equal(aux["unass"], z["unass"]), f


def test_oracle_mt_equals_sequential():
    x, y = _clouds(2, 1024, 9)
    d0, a0 = oracle.emd_forward(x, y, 0.005, 8)
    d1, a1 = oracle.emd_forward(x, y, 0.005, 8, mt=True)
    assert np.array_equal(a0, a1) and np.array_equal(d0, d1)


def test_oracle_properties_vs_hungarian():
    """Idea of the reference's own (commented out) test_emd, emd_module.py:98-118:
    dist is re-derivable from assignment; the auction's mean cost is close to the
    optimum (it may be lower: the forced last round is not a bijection)."""
    from scipy.optimize import linear_sum_assignment

    x, y = _clouds(1, 1024, 21)
    d, a = oracle.emd_forward(x, y, 0.005, 50)
    rec = ((x[0] - y[0][a[0]]) ** 2).sum(-1)
    np.testing.assert_allclose(d[0], rec, rtol=1e-5, atol=1e-9)
    cost = np.sqrt(((x[0][:, None, :] - y[0][None, :, :]) ** 2).sum(-1))
    r, c = linear_sum_assignment(cost)
    opt = cost[r, c].mean()
    got = np.sqrt(d[0]).mean()
    assert got < opt * 1.10, (got, opt)
    assert len(np.unique(a[0])) > 0.9 * 1024


def test_oracle_backward_formula():
    x, y = _clouds(2, 1024, 4)
    d, a = oracle.emd_forward(x, y, 0.005, 3)
    gd = np.random.default_rng(0).random((2, 1024), dtype=np.float32)
    g = oracle.emd_backward(x, y, gd, a)
    sel = np.take_along_axis(y, a[..., None].astype(np.int64).repeat(3, -1), 1)
    np.testing.assert_array_equal(g, (gd * 2)[..., None] * (x - sel))


def test_host_wrapper_asserts_like_reference():
    from sparenet_amd.cuda.emd.emd_module import emdModule

    with pytest.raises(AssertionError):
        emdModule()(torch.rand(1, 1000, 3), torch.rand(1, 1000, 3), 0.005, 5)
    with pytest.raises(AssertionError):
        emdModule()(torch.rand(1, 1024, 3), torch.rand(1, 2048, 3), 0.005, 5)


# ------------------------------------------------------------------ GPU side
def _hip(x, y, eps, iters, dev, stats=False):
    from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

    st = torch.zeros(2, dtype=torch.int64, device=dev) if stats else None
    d, a = emd_forward_raw(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), eps, iters, st)
    out = (d.cpu().numpy(), a.cpu().numpy())
    return out + (st.cpu().numpy(),) if stats else out


@pytest.mark.gpu
def test_hip_matches_golden(golden_dir, dev):
    for f in _golden(golden_dir):
        z = np.load(f)
        d, a, st = _hip(z["xyz1"], z["xyz2"], float(z["eps"]), int(z["iters"]), dev, stats=True)
        assert np.array_equal(a, z["assignment"]), f
        assert np.array_equal(d, z["dist"]), f
        assert st[0] == int(z["unass"].astype(np.int64).sum()) * z["xyz1"].shape[1], f


@pytest.mark.gpu
@pytest.mark.parametrize("skip,spread,scan", [("0", "0", None), ("2", "2", None), ("1", "3", None), ("2", "0", "1000000"),
                                              ("0", "2", "0")])
def test_hip_forced_paths_match_emulated_reference_golden(skip, spread, scan, golden_dir, dev, monkeypatch):
    """Round 5's data-dependent auction paths against the REFERENCE'S kernel text (the emulated goldens, which include
    contested geometries: targets on a sphere, bidders scattered through the cube around it, eps > 0 and < 0), not
    only against oracle/emd.c: the outbid-skip (SN_EMD_SKIP 0 = never, 2 = every iteration), the transposed rank split
    (SN_EMD_SPREAD 0 / 2 / 3) and both bid forms (SN_EMD_SCAN 0 = matrix-core search only, 10^6 = scan from the
    first iteration), each forced on and off."""
    monkeypatch.setenv("SN_EMD_SKIP", skip)
    monkeypatch.setenv("SN_EMD_SPREAD", spread)
    if scan is not None:
        monkeypatch.setenv("SN_EMD_SCAN", scan)
    seen_contested = 0
    for f in _golden(golden_dir):
        z = np.load(f)
        seen_contested += "contested" in os.path.basename(f)
        d, a, st = _hip(z["xyz1"], z["xyz2"], float(z["eps"]), int(z["iters"]), dev, stats=True)
        assert np.array_equal(a, z["assignment"]), (f, skip, spread, scan)
        assert np.array_equal(d, z["dist"]), (f, skip, spread, scan)
        assert st[0] == int(z["unass"].astype(np.int64).sum()) * z["xyz1"].shape[1], (f, skip, spread, scan)
    assert seen_contested >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,iters,eps,kind,seed", [
    (3, 1024, 7, 0.005, "uniform", 1),
    (2, 2048, 12, 0.002, "uniform", 2),
    (1, 4096, 5, 0.005, "uniform", 3),
    (2, 1024, 15, 0.005, "near", 4),
    (2, 1024, 6, 0.005, "lattice", 5),    # massive exact ties in the bid values
    (1, 3072, 4, 0.01, "lattice", 6),     # ties + two reference tiles of different delta
    (5, 1024, 1, 0.005, "uniform", 7),
    (1, 1024, 0, 0.005, "uniform", 8),    # iters = 0
    (9, 1024, 20, 0.005, "uniform", 9),   # batch not a multiple of 8 (XCD map), tail iterations
    (2, 2048, 10, -0.001, "uniform", 10), # negative eps: prices may fall (price floor of the filter)
    (1, 8192, 6, 0.005, "clustered", 11), # tight clusters: long hit queues, many exact batches
    (2, 1024, 8, 0.005, "far", 12),       # large offsets: |t|^2 - 2 t.x cancels heavily
])
def test_hip_matches_oracle(b, n, iters, eps, kind, seed, dev):
    x, y = _clouds(b, n, seed, kind)
    d0, a0 = oracle.emd_forward(x, y, eps, iters, mt=True)
    d1, a1 = _hip(x, y, eps, iters, dev)
    if iters == 0:
        assert (a1 == -1).all() and (d1 == 0).all()
        return
    assert np.array_equal(a0, a1)
    assert np.array_equal(d0, d1)


@pytest.mark.gpu
@pytest.mark.parametrize("scan", ["0", "48", "1000000"])
def test_hip_both_bid_paths_match_oracle(scan, dev, monkeypatch):
    """The bid phase has two forms: the matrix-core search of a group of 64 bidders (dense iterations) and the
    per-bidder scan by quarter waves (iterations with at most SN_EMD_SCAN bidders per workgroup, default 256).
    SN_EMD_SCAN is read per call: 0 = group search only, 48 = a switch late in the call (and the 2 / 4 quarters per
    bidder forms of the scan), 10^6 = the scan from the first iteration on (several passes of 64 bidders, no previous
    favourites for a start).  Ties, negative eps, clusters, far offsets, a size whose boxes do not fit the LDS copy."""
    monkeypatch.setenv("SN_EMD_SCAN", scan)
    for b, n, iters, eps, kind, seed in [(2, 1024, 15, 0.005, "near", 4), (2, 1024, 6, 0.005, "lattice", 5),
                                         (1, 3072, 4, 0.01, "lattice", 6), (2, 2048, 10, -0.001, "uniform", 10),
                                         (1, 8192, 6, 0.005, "clustered", 11), (2, 1024, 8, 0.005, "far", 12),
                                         (9, 1024, 20, 0.005, "uniform", 9), (1, 16384, 9, 0.005, "uniform", 13),
                                         (1, 32768, 3, 0.005, "uniform", 14)]:
        x, y = _clouds(b, n, seed, kind)
        d0, a0 = oracle.emd_forward(x, y, eps, iters, mt=True)
        d1, a1 = _hip(x, y, eps, iters, dev)
        assert np.array_equal(a0, a1), (scan, b, n, iters, kind)
        assert np.array_equal(d0, d1), (scan, b, n, iters, kind)


def _contested(b, n, seed, spread=1.0):
    """Targets on a sphere of radius 0.5, bidders scattered through a cube of half-width `spread` around it: the bidders
    outside the sphere all prefer the near-side targets -- hundreds of bidders per target, prices that keep climbing,
    an auction that does not converge in 50 iterations (what the refine stages of an untrained generator produce)."""
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(b, n, 3, generator=g)
    y = 0.5 * y / y.norm(dim=2, keepdim=True)
    x = y + spread * (2 * torch.rand(b, n, 3, generator=g) - 1)
    return x.contiguous().numpy(), y.contiguous().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("skip,spread", [("1", "1"), ("2", "2"), ("0", "3"), ("2", "0")])
def test_hip_contested_auction_paths_match_oracle(skip, spread, dev, monkeypatch):
    """Round 5's data-dependent paths, each forced on and off (read per call): the outbid-skip (SN_EMD_SKIP: a bidder
    that finds its target's running maximum already above its own increment + 1e-6 does not link itself; 2 = in every
    iteration, 1 = when the previous iteration's lists say so), the transposed rank split of scan iterations
    (SN_EMD_SPREAD: 2 = every iteration, 3 = every scan iteration, 1 = off-surface / contested clouds) and, through
    them, the scan for every workgroup of a contested iteration.  Contested clouds (lists of hundreds), ties, negative
    eps (no skip: prices fall, the persistent max_idx can decide), one iteration (the forced last one), team
    geometries from one workgroup per cloud (b = 40) to a whole XCD (b = 2)."""
    monkeypatch.setenv("SN_EMD_SKIP", skip)
    monkeypatch.setenv("SN_EMD_SPREAD", spread)
    cases = [("contested", 2, 2048, 12, 0.005, 21), ("contested", 9, 1024, 25, 0.005, 22),
             ("contested", 40, 1024, 6, 0.002, 23), ("contested", 1, 4096, 50, 0.005, 24),
             ("contested", 3, 1024, 1, 0.005, 25), ("contested", 2, 1024, 9, -0.001, 26),
             ("lattice", 2, 1024, 6, 0.005, 5), ("near", 2, 1024, 15, 0.005, 4), ("uniform", 9, 1024, 20, 0.005, 9),
             ("clustered", 1, 8192, 6, 0.005, 11), ("contested", 1, 3072, 8, 0.005, 27)]
    for kind, b, n, iters, eps, seed in cases:
        x, y = _contested(b, n, seed) if kind == "contested" else _clouds(b, n, seed, kind)
        d0, a0, aux = oracle.emd_forward(x, y, eps, iters, mt=True, return_aux=True)
        d1, a1, st = _hip(x, y, eps, iters, dev, stats=True)
        assert np.array_equal(a0, a1), (skip, spread, kind, b, n, iters)
        assert np.array_equal(d0, d1), (skip, spread, kind, b, n, iters)
        assert int(st[0]) == aux["pairs_eff"], (skip, spread, kind, b, n, iters)


@pytest.mark.gpu
def test_hip_prices_bit_exact_when_converged(dev):
    """Sensitive arithmetic check: when the auction converges before the last
    iteration (no forced assignment, hence no price race), the price vector --
    an accumulation of every winning bid increment, i.e. of sqrtf, the double
    detour and the top-2 logic -- must equal the oracle's bit for bit."""
    from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

    x, y = _clouds(2, 2048, 12, "near")
    d0, a0, aux = oracle.emd_forward(x, y, 0.005, 30, mt=True, return_aux=True)
    assert aux["unass"][-1] == 0, "pick a case that converges"
    d, a, ws = emd_forward_raw(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), 0.005, 30,
                               return_workspace=True)
    arr = (2 * 2048 * 4 + 255) // 256 * 256
    price = ws[arr:arr + 2 * 2048 * 4].view(torch.float32).view(2, 2048).cpu().numpy()
    assert np.array_equal(a.cpu().numpy(), a0)
    assert np.array_equal(price, aux["price"])


@pytest.mark.gpu
def test_hip_autograd_and_module_api(dev):
    from sparenet_amd.cuda.emd.emd_module import emdModule

    x, y = _clouds(2, 1024, 31)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    yt = torch.from_numpy(y).to(dev).requires_grad_(True)
    dist, assign = emdModule()(xt, yt, eps=0.005, iters=10)
    assert assign.dtype == torch.int32 and not assign.requires_grad
    loss = torch.sqrt(dist).mean(1).mean()
    loss.backward()
    d0, a0 = oracle.emd_forward(x, y, 0.005, 10)
    gd = (0.5 / np.sqrt(d0) / (2 * 1024)).astype(np.float32)
    g0 = oracle.emd_backward(x, y, gd, a0)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g0, rtol=1e-5, atol=1e-8)
    assert torch.count_nonzero(yt.grad) == 0


@pytest.mark.gpu
def test_hip_full_size_properties(dev):
    """BASELINE config 2: [32,16384,3], eps 0.005, 50 iterations."""
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(32, 16384, 3, generator=g).to(dev)
    y = torch.rand(32, 16384, 3, generator=g).to(dev)
    from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

    st = torch.zeros(2, dtype=torch.int64, device=dev)
    d, a = emd_forward_raw(x, y, 0.005, 50, st)
    assert int(a.min()) >= 0 and int(a.max()) < 16384
    sel = torch.gather(y, 1, a.long().unsqueeze(-1).expand(-1, -1, 3))
    diff = x - sel
    rec = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal(rec, d)
    # determinism: a second run is bit-identical
    d2, a2 = emd_forward_raw(x, y, 0.005, 50)
    assert torch.equal(a, a2) and torch.equal(d, d2)
    # most targets used exactly once, mean cost in the expected range for uniform clouds
    uniq = torch.stack([torch.unique(a[i]).numel() * 1.0 for i in range(32)] if False else
                       [torch.tensor(float(torch.unique(a[i]).numel())) for i in range(32)])
    assert float(uniq.mean()) > 0.93 * 16384
    m = float(torch.sqrt(d).mean())
    assert 0.01 < m < 0.05, m
    assert int(st[0]) >= 32 * 16384 * 16384


def _racy_golden(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "xfail_emd_*.npz")))


def test_oracle_matches_canonical_run_of_schedule_dependent_golden(golden_dir):
    """Fixtures on which the reference's OWN kernels race (several bidders inside GetMax's +-1e-6 window,
    emd_cuda.cu:188-191: the last writer wins) are kept as xfail_emd_*.npz with the CANONICAL run (threads started in
    order -- the sequential ascending-j order the oracle and the HIP kernel implement).  The oracle must reproduce
    that run; that other schedules give other results is the reference's race, recorded in the fixture."""
    files = _racy_golden(golden_dir)
    for f in files:
        z = np.load(f)
        assert not bool(z["schedule_invariant"])
        if "agrees_with_oracle" in z.files and not bool(z["agrees_with_oracle"]):
            continue   # a fixture the oracle's rule does not explain stays an expected failure (none today)
        d, a, aux = oracle.emd_forward(z["xyz1"], z["xyz2"], float(z["eps"]), int(z["iters"]), return_aux=True)
        assert np.array_equal(a, z["assignment"]) and np.array_equal(d, z["dist"]), f
        assert np.array_equal(aux["unass"], z["unass"]), f


@pytest.mark.gpu
def test_hip_matches_canonical_run_of_schedule_dependent_golden(golden_dir, dev):
    for f in _racy_golden(golden_dir):
        z = np.load(f)
        if "agrees_with_oracle" in z.files and not bool(z["agrees_with_oracle"]):
            continue
        d, a = _hip(z["xyz1"], z["xyz2"], float(z["eps"]), int(z["iters"]), dev)
        assert np.array_equal(a, z["assignment"]) and np.array_equal(d, z["dist"]), f
