"""oracle -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (sparenet_amd) never imports this module.

numpy in, numpy out.  The C sources next to this file are compiled into
liboracle.so by `make -C oracle` (also done by __graft_entry__.build()).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.isfile(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in ("oracle_emd_forward", "oracle_emd_forward_mt"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_longlong
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_f)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_i)


def _pf(a):
    return a.ctypes.data_as(c_f)


def _pi(a):
    return a.ctypes.data_as(c_i)


# --------------------------------------------------------------------- chamfer
def chamfer_forward(xyz1, xyz2, mt=False):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32)
    d2 = np.zeros((b, m), np.float32)
    i1 = np.zeros((b, n), np.int32)
    i2 = np.zeros((b, m), np.int32)
    fn = lib().oracle_chamfer_forward_mt if mt else lib().oracle_chamfer_forward
    fn(p1, p2, b, n, m, _pf(d1), _pi(i1), _pf(d2), _pi(i2))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, gd1, gd2, idx1, idx2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    gd1, pg1 = _f(gd1)
    gd2, pg2 = _f(gd2)
    idx1, pi1 = _i(idx1)
    idx2, pi2 = _i(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros_like(xyz1)
    g2 = np.zeros_like(xyz2)
    lib().oracle_chamfer_backward(p1, p2, pg1, pg2, pi1, pi2, b, n, m, _pf(g1), _pf(g2))
    return g1, g2


# ------------------------------------------------------------------------- emd
def emd_forward(xyz1, xyz2, eps, iters, mt=False, return_aux=False):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    assert xyz2.shape[1] == n
    dist = np.zeros((b, n), np.float32)
    assign = np.zeros((b, n), np.int32)
    price = np.zeros((b, n), np.float32)
    trace = np.zeros((max(iters, 1),), np.int32)
    fn = lib().oracle_emd_forward_mt if mt else lib().oracle_emd_forward
    pairs = fn(p1, p2, b, n, ctypes.c_float(eps), int(iters), _pf(dist), _pi(assign),
               _pf(price), _pi(trace))
    if return_aux:
        return dist, assign, dict(price=price, unass=trace[:iters], pairs_eff=int(pairs))
    return dist, assign


def emd_backward(xyz1, xyz2, graddist, assignment):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    gd, pg = _f(graddist)
    a, pa = _i(assignment)
    b, n, _ = xyz1.shape
    g = np.zeros_like(xyz1)
    lib().oracle_emd_backward(p1, p2, pg, pa, b, n, _pf(g))
    return g


# ------------------------------------------------------------------- expansion
def expansion_forward(xyz, primitive_size, alpha):
    xyz, p = _f(xyz)
    b, n, _ = xyz.shape
    dist = np.zeros((b, n), np.float32)
    assign = np.zeros((b, n), np.int32)
    mean = np.zeros((b,), np.float32)
    lib().oracle_expansion_forward(p, b, n, int(primitive_size), ctypes.c_float(alpha),
                                   _pf(dist), _pi(assign), _pf(mean))
    return dist, assign, mean


def expansion_backward(xyz, graddist, assignment):
    xyz, p = _f(xyz)
    gd, pg = _f(graddist)
    a, pa = _i(assignment)
    b, n, _ = xyz.shape
    g = np.zeros_like(xyz)
    lib().oracle_expansion_backward(p, pg, pa, b, n, _pf(g))
    return g


# ------------------------------------------------------------------------- mds
def mds(xyz, npoint, mean_mst_length, exp_mode=1, bs_override=0):
    xyz, p = _f(xyz)
    mml, pm = _f(mean_mst_length)
    b, n, _ = xyz.shape
    idx = np.zeros((b, npoint), np.int32)
    lib().oracle_mds(p, b, n, int(npoint), pm, int(exp_mode), int(bs_override), _pi(idx))
    return idx


def gather_forward(feat, idx):
    feat, pf = _f(feat)
    idx, pi = _i(idx)
    b, c, n = feat.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_gather_forward(pf, pi, b, c, n, m, _pf(out))
    return out


def gather_backward(grad_out, idx, n):
    go, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = go.shape
    gf = np.zeros((b, c, n), np.float32)
    lib().oracle_gather_backward(pg, pi, b, c, n, m, _pf(gf))
    return gf


# ------------------------------------------------------------------------- p2i
def p2i_max_forward(points, feat, batch_inds, background, radius, mt=False):
    points, pp = _f(points)
    feat, pf = _f(feat)
    bi, pb = _i(batch_inds)
    bg = np.ascontiguousarray(background, dtype=np.float32)
    B, C, H, W = bg.shape
    n = points.shape[0]
    out = bg.copy()
    ids = np.full((B, C, H, W), -1, np.int32)
    fn = lib().oracle_p2i_max_forward_mt if mt else lib().oracle_p2i_max_forward
    fn(pp, pf, pb, n, C, B, H, W, ctypes.c_float(radius), _pf(out), _pi(ids))
    return out, ids


def p2i_max_backward(out_grad, out_ids, points, feat, radius):
    og, pg = _f(out_grad)
    ids, pi = _i(out_ids)
    points, pp = _f(points)
    feat, pf = _f(feat)
    B, C, H, W = og.shape
    n = points.shape[0]
    gp = np.zeros_like(points)
    gf = np.zeros_like(feat)
    gb = np.zeros_like(og)
    lib().oracle_p2i_max_backward(pg, pi, pp, pf, n, C, B, H, W, ctypes.c_float(radius),
                                  _pf(gp), _pf(gf), _pf(gb))
    return gp, gf, gb


def p2i_max_backward_exact(out_grad, out_ids, points, feat, radius):
    """(points_grad, feat_grad): the fp32 terms of p2i_max_backward summed exactly (in double), rounded once."""
    og, pg = _f(out_grad)
    ids, pi = _i(out_ids)
    points, pp = _f(points)
    feat, pf = _f(feat)
    B, C, H, W = og.shape
    gp = np.zeros_like(points)
    gf = np.zeros_like(feat)
    lib().oracle_p2i_max_backward_exact(pg, pi, pp, pf, points.shape[0], C, B, H, W, ctypes.c_float(radius),
                                        _pf(gp), _pf(gf))
    return gp, gf


def p2i_sum_forward(points, feat, batch_inds, background, radius):
    points, pp = _f(points)
    feat, pf = _f(feat)
    bi, pb = _i(batch_inds)
    bg = np.ascontiguousarray(background, dtype=np.float32)
    B, C, H, W = bg.shape
    out = bg.copy()
    lib().oracle_p2i_sum_forward(pp, pf, pb, points.shape[0], C, B, H, W,
                                 ctypes.c_float(radius), _pf(out))
    return out


def p2i_sum_backward(out_grad, points, feat, batch_inds, radius):
    og, pg = _f(out_grad)
    points, pp = _f(points)
    feat, pf = _f(feat)
    bi, pb = _i(batch_inds)
    B, C, H, W = og.shape
    gp = np.zeros_like(points)
    gf = np.zeros_like(feat)
    lib().oracle_p2i_sum_backward(pg, pp, pf, pb, points.shape[0], C, B, H, W,
                                  ctypes.c_float(radius), _pf(gp), _pf(gf))
    return gp, gf


# -------------------------------------------------------------------- gridding
def gridding_forward(ptcloud, scale):
    """ptcloud [B,n,3] already multiplied by scale//2 (as cuda/gridding/__init__.py:45-48)."""
    pc, pp = _f(ptcloud)
    b, n, _ = pc.shape
    grid = np.zeros((b, scale ** 3), np.float32)
    w = np.zeros((b, n, 8, 3), np.float32)
    ix = np.zeros((b, n, 8), np.int32)
    lib().oracle_gridding_forward(pp, b, n, int(scale), _pf(grid), _pf(w), _pi(ix))
    return grid, w, ix


def gridding_dist_forward(ptcloud, bounds):
    """ptcloud [B,n,3] (already scaled), bounds = (min_x, max_x, min_y, max_y, min_z, max_z) integers
    -> grid [B, nverts, 8], weights [B,n,8,3], indexes [B,n,8] (cuda/gridding_loss)."""
    pc, pp = _f(ptcloud)
    b, n, _ = pc.shape
    mnx, mxx, mny, mxy, mnz, mxz = [int(v) for v in bounds]
    nverts = (mxx - mnx + 1) * (mxy - mny + 1) * (mxz - mnz + 1)
    grid = np.zeros((b, nverts, 8), np.float32)
    w = np.zeros((b, n, 8, 3), np.float32)
    ix = np.zeros((b, n, 8), np.int32)
    lib().oracle_gridding_dist_forward(pp, b, n, mnx, mxx, mny, mxy, mnz, mxz, _pf(grid), _pf(w), _pi(ix))
    return grid, w, ix


def gridding_backward(grad_grid, weights, indexes):
    gg, pg = _f(grad_grid)
    w, pw = _f(weights)
    ix, pi = _i(indexes)
    b, n = ix.shape[:2]
    out = np.zeros((b, n, 3), np.float32)
    lib().oracle_gridding_backward(pg, pw, pi, b, n, gg.shape[1], _pf(out))
    return out


def gridding_reverse_forward(grid, scale):
    g, pg = _f(grid)
    b = g.shape[0]
    out = np.zeros((b, scale ** 3, 3), np.float32)
    lib().oracle_gridding_reverse_forward(pg, b, int(scale), _pf(out))
    return out


def gridding_reverse_backward(grad_ptcloud, grid, ptcloud, scale):
    """ptcloud = raw forward output (before the module's / scale * 2)."""
    gp, pgp = _f(grad_ptcloud)
    g, pg = _f(grid)
    pc, ppc = _f(ptcloud)
    b = g.shape[0]
    out = np.zeros((b, scale, scale, scale), np.float32)
    lib().oracle_gridding_reverse_backward(pgp, pg, ppc, b, int(scale), _pf(out))
    return out


def cubic_forward(ptcloud, feat, ns):
    """ptcloud [B,n,3] already mapped to voxel space (p*h+h)."""
    pc, pp = _f(ptcloud)
    f, pf = _f(feat)
    b, n, _ = pc.shape
    c, scale = f.shape[1], f.shape[2]
    nv = (2 * ns) ** 3
    out = np.zeros((b, n, nv, c), np.float32)
    idx = np.zeros((b, n, nv), np.int32)
    lib().oracle_cubic_forward(pp, pf, b, n, c, scale, int(ns), _pf(out), _pi(idx))
    return out, idx


def cubic_backward(grad_out, idx, c, scale, ns):
    go, pg = _f(grad_out)
    ix, pi = _i(idx)
    b, n = ix.shape[:2]
    out = np.zeros((b, c, scale, scale, scale), np.float32)
    lib().oracle_cubic_backward(pg, pi, b, n, c, scale, int(ns), _pf(out))
    return out


# -------------------------------------------------------------------- k-NN graph (numpy only)
def knn(x, k):
    """x [B,C,N] -> idx [B,N,k]: the reference's ranking  |x_j|^2 - 2 x_i.x_j  (the CPU branch of
    models/sparenet_generator.py:872-875 up to the row constant) in float64, ascending, equal scores by
    lower index."""
    x = np.asarray(x, np.float64)
    inner = np.einsum("bci,bcj->bij", x, x)
    score = (x * x).sum(1)[:, None, :] - 2.0 * inner
    return np.argsort(score, axis=2, kind="stable")[:, :, :k]


def graph_feature(x, idx):
    """x [B,C,N], idx [B,N,k] -> [B,2C,N,k] (models/sparenet_generator.py:880-906)."""
    x = np.asarray(x, np.float32)
    b, c, n = x.shape
    k = idx.shape[2]
    nb = np.take_along_axis(x[:, :, None, :].repeat(n, 2), idx[:, None, :, :].repeat(c, 1), axis=3)
    own = x[:, :, :, None].repeat(k, 3)
    return np.concatenate([nb - own, own], axis=1)
