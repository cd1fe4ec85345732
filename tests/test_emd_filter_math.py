"""The inequalities behind the auction's pruning, checked on the CPU in the kernel's own fp32 arithmetic.

`bid_scan` (and, block by block, `bid_group`) evaluates a target only if (1) the box of its 16-target block lies
within the bidder's REACH, r2 = coarse_threshold(cm, slack, a_max) x 1.0001, and (2) it passes the precise filter
s <= r |r|, r = filter_target(price) - filter_thr(cm), where cm is the smaller value of two previous favourites (any
two distinct targets) at today's prices.  The claim that makes this exact: every target whose value reaches cm
survives both tests -- so the top two values of the survivors are the top two of the full scan.  The formulas
below restate sparenet_amd/csrc/emd.hip (filter_target, filter_thr, filter_pass, coarse_threshold, box_within,
bid_value) operation by operation in numpy float32; the clouds include large offsets (heavy cancellation in
|t|^2 - 2 t.x), lattices (exact ties), tight clusters and prices of very different sizes."""
import zlib

import numpy as np
import pytest

f32 = np.float32
EPS20 = f32(9.5367431640625e-07)   # 2^-20


def filter_target(p):
    return (f32(3.0) - p) + (f32(3.0) + np.abs(p)) * EPS20


def filter_thr(c):
    return c - (f32(3.0) + np.abs(c)) * EPS20


def fmaf(a, b, c):
    return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)


def coarse_threshold(cm, base, a_max):
    r = a_max - filter_thr(cm)
    return fmaf(r * np.abs(r), f32(1.00000095367431640625), base)


def sq_dist(t, x):
    d = t - x
    xx, yy, zz = d[..., 0] * d[..., 0], d[..., 1] * d[..., 1], d[..., 2] * d[..., 2]
    return (xx + yy) + zz


def bid_value(t, p, x):
    s = sq_dist(t, x)
    return ((3.0 - np.sqrt(s).astype(np.float64)) - p.astype(np.float64)).astype(np.float32)


def clouds(kind, n, rng):
    x = rng.random((n, 3), dtype=np.float32)
    y = rng.random((n, 3), dtype=np.float32)
    if kind == "far":
        x, y = x + f32(50.0), y + f32(50.0)
    elif kind == "lattice":
        x = (rng.integers(0, 8, (n, 3)) / 7.0).astype(np.float32)
        y = (rng.integers(0, 8, (n, 3)) / 7.0).astype(np.float32)
    elif kind == "clustered":
        c = rng.random((6, 3), dtype=np.float32)
        x = (c[rng.integers(0, 6, n)] + f32(0.003) * rng.standard_normal((n, 3)).astype(np.float32)).clip(0, 1)
        y = (c[rng.integers(0, 6, n)] + f32(0.003) * rng.standard_normal((n, 3)).astype(np.float32)).clip(0, 1)
    elif kind == "negative":
        x, y = x - f32(3.0), y - f32(3.0)
    return x.astype(np.float32), y.astype(np.float32)


@pytest.mark.parametrize("kind", ["uniform", "far", "lattice", "clustered", "negative"])
@pytest.mark.parametrize("price_scale", [0.0, 0.02, 0.5])
def test_no_target_that_matters_is_pruned(kind, price_scale):
    rng = np.random.default_rng(zlib.crc32(f"{kind}/{price_scale}".encode()))
    n = 2048
    x, y = clouds(kind, n, rng)
    order = np.lexsort((y[:, 2], y[:, 1], y[:, 0]))   # any split into blocks of 16 neighbours-ish will do
    y = y[order]
    price = (rng.random(n, dtype=np.float32) * f32(price_scale)).astype(np.float32)
    if price_scale:
        price[rng.integers(0, n, n // 8)] = 0.0        # unassigned targets keep price 0
    blk = y.reshape(n // 16, 16, 3)
    lo, hi = blk.min(1), blk.max(1)
    box_lo, box_hi = y.min(0), y.max(0)
    tmax = f32(0.0)
    for a in range(3):
        tmax = tmax + np.maximum(box_lo[a] * box_lo[a], box_hi[a] * box_hi[a])
    tmax = tmax * f32(1.0001)
    a_max = filter_target(f32(0.0)) + EPS20
    pruned_blocks = pruned_targets = 0
    for j in rng.integers(0, n, 300):
        xb = x[j]
        val = bid_value(y, price, xb)
        # two favourites: a near pair, a random pair, or the true top two with their prices raised afterwards
        mode = j % 3
        if mode == 0:
            pa, pb = np.argsort(sq_dist(y, xb))[:2]
        elif mode == 1:
            pa, pb = rng.choice(n, 2, replace=False)
        else:
            pa, pb = np.argsort(-val)[:2]
        cm = np.minimum(val[pa], val[pb])
        xx = (xb[0] * xb[0] + xb[1] * xb[1]) + xb[2] * xb[2]
        v = coarse_threshold(cm, f32(2.0) * f32(3.814697265625e-06) * (tmax + xx), a_max)
        r2 = v * f32(1.0001) if v > 0 else v
        g = np.maximum(np.maximum(lo - xb, xb - hi), f32(0.0))
        within = ((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]) * f32(0.9999) <= r2
        s = sq_dist(y, xb)
        r = filter_target(price) - filter_thr(cm)
        passes = s <= r * np.abs(r)
        kept = np.repeat(within, 16) & passes
        matters = val >= cm
        assert not (matters & ~kept).any(), (kind, price_scale, int(j), np.flatnonzero(matters & ~kept)[:4])
        # and therefore the survivors' top two are the full scan's
        top = np.sort(val)[-2:]
        assert kept.sum() >= 2 and np.array_equal(np.sort(val[kept])[-2:], top)
        pruned_blocks += int((~within).sum())
        pruned_targets += int((~kept).sum())
    if kind in ("uniform", "clustered") and price_scale <= 0.02:   # the tests are not vacuous: most of the cloud is pruned
        assert pruned_blocks > 0.4 * 300 * (n // 16) and pruned_targets > 0.6 * 300 * n
