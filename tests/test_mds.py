"""Minimum density sampling + gather.

CPU: oracle (libm expf, bs=1) vs golden index sequences from the reference kernel's
race-free single-thread instantiation (tests/golden/gen_emulated.py mds); the cross-
thread tie rule vs a direct simulation of the reference's reduction tree
(MDS_cuda.cu:81-87, :139-198); sn_expf vs libm.
GPU: HIP vs oracle (sn_expf, reference thread count): index sequences exact.
"""
import glob
import os

import numpy as np
import pytest
import torch

import oracle


def _golden(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "mds_*.npz")))
    assert files
    return files


def test_oracle_matches_emulated_single_thread_kernel(golden_dir):
    for f in _golden(golden_dir):
        z = np.load(f)
        if z["xyz"].shape[1] > 10000:
            continue  # the 19384-point case runs on the GPU side (keeps the CPU suite short)
        idx = oracle.mds(z["xyz"], int(z["npoint"]), z["mean_mst_length"], exp_mode=0, bs_override=1)
        assert np.array_equal(idx, z["idx_bs1"]), f


def _tree_winner(vals, bs):
    """Direct simulation of the reference block reduction: __update keeps the lower slot
    on ties (MDS_cuda.cu:81-87), strides bs/2 ... 1 (:139-198)."""
    cnt = list(vals)
    cnt_i = list(range(bs))
    s = bs // 2
    while s >= 1:
        for t in range(s):
            v1, v2 = cnt[t], cnt[t + s]
            i1, i2 = cnt_i[t], cnt_i[t + s]
            cnt[t] = min(v1, v2)
            cnt_i[t] = i2 if v2 < v1 else i1
        s //= 2
    return cnt_i[0]


def test_tie_rule_is_bit_reversed_thread_order():
    rng = np.random.default_rng(0)
    for bs in (8, 16, 64, 256, 1024):
        lg = bs.bit_length() - 1
        for _ in range(20):
            vals = rng.integers(0, 3, bs).astype(np.float32)  # many ties
            w = _tree_winner(vals, bs)
            mn = vals.min()
            tied = [t for t in range(bs) if vals[t] == mn]
            rev = lambda t: int(format(t, f"0{lg}b")[::-1], 2)
            assert w == min(tied, key=rev)


def test_oracle_tie_order_with_exact_zero_densities():
    """With a tiny mean_mst_length exp(-d/t) underflows to exactly 0 for every other point,
    so all densities tie at 0 and the pick order is purely the tie rule:
    argmin (bitrev(k mod bs), k)."""
    n, m = 64, 10
    rng = np.random.default_rng(1)
    x = rng.random((1, n, 3), dtype=np.float32)
    idx = oracle.mds(x, m, np.array([1e-4], np.float32), exp_mode=1)
    rev = lambda t: int(format(t, "06b")[::-1], 2)
    order = sorted(range(1, n), key=lambda k: (rev(k % 64), k))
    assert idx[0, 0] == 0 and list(idx[0, 1:]) == order[:m - 1]


def test_oracle_rows_are_unique_and_greedy_in_float64():
    rng = np.random.default_rng(2)
    x = rng.random((2, 500, 3), dtype=np.float32)
    mml = np.array([0.05, 0.08], np.float32)
    idx = oracle.mds(x, 200, mml, exp_mode=1)
    for b in range(2):
        assert len(set(idx[b])) == 200 and idx[b, 0] == 0
        # greedy-min property re-checked in float64 with a tolerance
        t = 5.0 * float(mml[b]) ** 2
        dens = np.zeros(500)
        taken = np.zeros(500, bool)
        taken[0] = True
        last = 0
        for j in range(1, 200):
            d = ((x[b].astype(np.float64) - x[b, last]) ** 2).sum(-1)
            dens += np.exp(-d / t)
            pick = idx[b, j]
            assert not taken[pick]
            assert dens[pick] <= dens[~taken].min() * (1 + 1e-4) + 1e-30
            taken[pick] = True
            last = pick


def test_sn_expf_close_to_libm_and_modes_agree_on_easy_case():
    rng = np.random.default_rng(3)
    x = rng.random((1, 200, 3), dtype=np.float32)
    mml = np.array([0.3], np.float32)  # wide kernel: no underflow, no ties
    a = oracle.mds(x, 50, mml, exp_mode=0)
    b = oracle.mds(x, 50, mml, exp_mode=1)
    assert (a == b).mean() > 0.9  # 1-ulp exp differences may flip a near-tie eventually


def test_gather_oracle():
    rng = np.random.default_rng(4)
    f = rng.random((2, 4, 50), dtype=np.float32)
    idx = np.stack([rng.permutation(50)[:20] for _ in range(2)]).astype(np.int32)
    out = oracle.gather_forward(f, idx)
    assert np.array_equal(out, np.take_along_axis(f, idx[:, None, :].repeat(4, 1), 2))
    g = rng.random((2, 4, 20), dtype=np.float32)
    gf = oracle.gather_backward(g, idx, 50)
    ref = np.zeros_like(f)
    for b in range(2):
        ref[b][:, idx[b]] = g[b]
    assert np.array_equal(gf, ref)


# ------------------------------------------------------------------ GPU side
def _hip_mds(x, m, mml, dev):
    from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample

    return minimum_density_sample(torch.from_numpy(x).to(dev), m, torch.from_numpy(mml).to(dev)).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,mml", [
    (2, 300, 128, 0.05), (1, 9000, 600, 0.012), (3, 64, 64, 0.2), (2, 1024, 512, 0.03),
    (1, 2048, 300, 1e-4),      # all-zero densities: pure tie-rule order, bs = 1024
    (2, 5000, 1000, 0.02), (1, 19384, 1500, 0.008), (1, 17, 9, 0.1), (1, 1, 1, 0.1),
    (1, 22000, 400, 0.01),     # 22 points per lane: z-in-LDS variant
    (1, 30000, 300, 0.01),     # generic fallback with the state in global memory
])
def test_hip_matches_oracle(b, n, m, mml, dev):
    rng = np.random.default_rng(n + m)
    x = rng.random((b, n, 3), dtype=np.float32)
    mm = (mml * (1 + 0.1 * rng.random(b))).astype(np.float32)
    ref = oracle.mds(x, m, mm, exp_mode=1)
    got = _hip_mds(x, m, mm, dev)
    assert np.array_equal(got, ref)


@pytest.mark.gpu
def test_hip_gather_fwd_bwd(dev):
    from sparenet_amd.cuda.MDS.MDS_module import gather_operation

    rng = np.random.default_rng(9)
    f = rng.random((3, 4, 1000), dtype=np.float32)
    idx = np.stack([rng.permutation(1000)[:700] for _ in range(3)]).astype(np.int32)
    ft = torch.from_numpy(f).to(dev).requires_grad_(True)
    out = gather_operation(ft, torch.from_numpy(idx).to(dev))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.gather_forward(f, idx))
    g = rng.random((3, 4, 700), dtype=np.float32)
    (out * torch.from_numpy(g).to(dev)).sum().backward()
    assert np.array_equal(ft.grad.cpu().numpy(), oracle.gather_backward(g, idx, 1000))


@pytest.mark.gpu
def test_hip_full_size_sparenet_shape(golden_dir, dev):
    """SpareNet call shape (models/sparenet_generator.py:568-573): n = 16384+3000, m = 16384,
    B = 4 here; one cloud is checked index-exactly against the oracle.  Also closes the
    chain for the 19384-point golden: oracle(libm, bs=1) == reference kernel<1> there."""
    rng = np.random.default_rng(1234)
    x = rng.random((4, 19384, 3), dtype=np.float32)
    mm = np.full(4, 0.0085, np.float32)
    got = _hip_mds(x, 16384, mm, dev)
    for b in range(4):
        assert len(set(got[b])) == 16384 and got[b, 0] == 0
    ref = oracle.mds(x[:1], 16384, mm[:1], exp_mode=1)
    assert np.array_equal(got[:1], ref)
    z = np.load(os.path.join(golden_dir, "mds_1x19384_m1024.npz"))
    idx = oracle.mds(z["xyz"], int(z["npoint"]), z["mean_mst_length"], exp_mode=0, bs_override=1)
    assert np.array_equal(idx, z["idx_bs1"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mml", [("sphere", 0.010), ("patches", 0.015)])
def test_hip_full_length_on_surface_clouds(kind, mml, dev):
    """All 16384 picks of a 19384-point SURFACE cloud (SpareNet's refine stage) against the oracle: late in
    such a run every region already holds picks, which is where the absorption culling (an increment below
    half an ulp of every density of a slot is skipped) prunes most -- it must never change a pick."""
    rng = np.random.default_rng(5)
    n, m = 19384, 16384
    if kind == "sphere":
        v = rng.standard_normal((1, n, 3)).astype(np.float32)
        x = (0.5 * v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32)
    else:
        c = rng.standard_normal((1, 32, 3)).astype(np.float32)
        c = 0.5 * c / np.linalg.norm(c, axis=2, keepdims=True)
        pick = np.take_along_axis(c, rng.integers(0, 32, (1, n, 1)).repeat(3, 2), 1)
        x = (pick + 0.03 * rng.standard_normal((1, n, 3)).astype(np.float32)).astype(np.float32)
    mm = np.array([mml], np.float32)
    assert np.array_equal(_hip_mds(x, m, mm, dev), oracle.mds(x, m, mm, exp_mode=1))


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m", [(1, 19384, 3000), (5, 19384, 2500), (12, 6000, 3000), (33, 4096, 2000),
                                   (70, 2048, 1500)])
def test_hip_dense_regime_teams_and_mixed_batches(b, n, m, dev):
    """Clouds whose cut ball covers a good part of their bounding box are sampled by a TEAM of workgroups
    (mds_dense_team_kernel: 16 / 8 / 4 / 2 workgroups per cloud depending on the batch, candidates exchanged through
    stamped 64-bit words), the others by the one-workgroup kernel -- in ONE call when a batch mixes both.  The batch
    here alternates mean MST lengths on both sides of the cross-over; every row must equal the oracle's sequence."""
    rng = np.random.default_rng(b * 1000 + n)
    x = rng.random((b, n, 3), dtype=np.float32)
    mm = np.where(np.arange(b) % 2 == 0, 0.09, 0.012).astype(np.float32) * (1 + 0.2 * rng.random(b, dtype=np.float32))
    got = _hip_mds(x, m, mm, dev)
    want = oracle.mds(x, m, mm, exp_mode=1)
    assert np.array_equal(got, want)
    got2 = _hip_mds(x, m, mm, dev)          # run to run: the exchange is deterministic
    assert np.array_equal(got, got2)


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,mml,kind", [
    (4, 2048, 2048, 0.2, "uniform"),        # m == n on a team: the last exchanges find nothing below 1e9 but one point
    (2, 8192, 8192, 0.004, "uniform"),      # m == n, surface regime, teams of 32: the pick cap (16 per exchange) saturates
    (3, 19384, 6000, 0.05, "duplicates"),   # exact duplicates: equal densities, the tie key decides inside the replay
    (8, 19384, 3000, 0.0085, "sphere"),     # teams of 32 in the surface regime (every cloud on a team since round 6)
    (9, 19384, 2000, 0.03, "uniform"),      # 9 clouds: two teams per XCD, teams of 16
    (1, 8192, 4000, 1e-4, "uniform"),       # all-zero densities: the picks are the tie rule's order, candidate by candidate
])
def test_hip_team_multi_pick_corner_cases(b, n, m, mml, kind, dev):
    """Round 6's team kernel: several exact picks per exchange (each member's lowest candidate with its coordinates +
    its second-lowest density; the replay accepts a pick only strictly below every member's second-lowest), members
    owning evenly dealt groups of 64 points, teams of up to 32.  Index-exact against the oracle where a replay error
    would show: everything selected, duplicated points, ties, both regimes, several team geometries."""
    rng = np.random.default_rng(n * 7 + m + b)
    if kind == "sphere":
        v = rng.standard_normal((b, n, 3)).astype(np.float32)
        x = (0.5 * v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32)
    else:
        x = rng.random((b, n, 3), dtype=np.float32)
    if kind == "duplicates":
        x[:, n // 2:] = x[:, :n - n // 2]     # every point twice
    mm = (mml * (1 + 0.1 * rng.random(b))).astype(np.float32)
    want = oracle.mds(x, m, mm, exp_mode=1)
    got = _hip_mds(x, m, mm, dev)
    assert np.array_equal(got, want)
    if m == n:
        assert all(len(set(r.tolist())) == n for r in got)
