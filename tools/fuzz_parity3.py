"""Randomised parity sweep, part 3 (by hand on a GPU box): the backward passes and the GRNet / EdgeConv ops
against the oracle at random shapes.  `python tools/fuzz_parity3.py [seconds] [seed]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from sparenet_amd.cuda.chamfer_distance import ChamferDistanceFunction
from sparenet_amd.cuda.emd.emd_module import emdFunction
from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyFunction
from sparenet_amd.cuda.MDS.MDS_module import gather_operation
from sparenet_amd.cuda.p2i_op import ext
from sparenet_amd.cuda.gridding import GriddingFunction, GriddingReverseFunction
from sparenet_amd.cuda.cubic_feature_sampling import CubicFeatureSamplingFunction
from sparenet_amd.cuda.knn import knn_fused, knn_unfused, get_graph_feature

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fails = {}
runs = 0


def check(name, ok, info):
    if not ok:
        fails[name] = fails.get(name, 0) + 1
        print("MISMATCH", name, info)


t_end = time.time() + budget
while time.time() < t_end:
    runs += 1
    # ---- Chamfer forward + backward (bit-exact own terms, scatter terms within fp32 atomics)
    b, n, m = int(rng.integers(1, 4)), int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    x, y = rng.random((b, n, 3), dtype=np.float32), rng.random((b, m, 3), dtype=np.float32)
    xt, yt = T(x).requires_grad_(True), T(y).requires_grad_(True)
    d1, d2 = ChamferDistanceFunction.apply(xt, yt)
    g1, g2 = rng.random((b, n), dtype=np.float32), rng.random((b, m), dtype=np.float32)
    ((d1 * T(g1)).sum() + (d2 * T(g2)).sum()).backward()
    r1, r2, i1, i2 = oracle.chamfer_forward(x, y, mt=True)
    rg1, rg2 = oracle.chamfer_backward(x, y, g1, g2, i1, i2)
    check("chamfer", np.array_equal(d1.detach().cpu().numpy(), r1) and np.array_equal(d2.detach().cpu().numpy(), r2)
          and np.array_equal(xt.grad.cpu().numpy(), rg1)
          and np.array_equal(yt.grad.cpu().numpy(), rg2), dict(b=b, n=n, m=m))
    # ---- EMD forward + backward (bit-exact: single writer per element)
    b, n = int(rng.integers(1, 4)), int(rng.choice([1024, 2048]))
    x, y = rng.random((b, n, 3), dtype=np.float32), rng.random((b, n, 3), dtype=np.float32)
    xt = T(x).requires_grad_(True)
    iters = int(rng.choice([1, 3, 10]))
    dist, assign = emdFunction.apply(xt, T(y), 0.005, iters)
    gd = rng.random((b, n), dtype=np.float32)
    (dist * T(gd)).sum().backward()
    d0, a0 = oracle.emd_forward(x, y, 0.005, iters, mt=True)
    check("emd", np.array_equal(assign.cpu().numpy(), a0) and np.array_equal(dist.detach().cpu().numpy(), d0)
          and np.array_equal(xt.grad.cpu().numpy(), oracle.emd_backward(x, y, gd, a0)), dict(b=b, n=n, iters=iters))
    # ---- expansion backward
    P = int(rng.choice([4, 32, 128, 512])); b = int(rng.integers(1, 3)); x = rng.random((b, P * int(rng.integers(1, 4)), 3), dtype=np.float32)
    xt = T(x).requires_grad_(True)
    d, a, _ = expansionPenaltyFunction.apply(xt, P, 1.2)
    gd = rng.random(d.shape, dtype=np.float32)
    (d * T(gd)).sum().backward()
    d0, a0, _ = oracle.expansion_forward(x, P, 1.2)
    check("expansion_bwd", np.array_equal(a.cpu().numpy(), a0)
          and np.array_equal(xt.grad.cpu().numpy(), oracle.expansion_backward(x, gd, a0)), dict(P=P, shape=x.shape))
    # ---- gather forward + backward
    b, c, n, m = int(rng.integers(1, 4)), int(rng.integers(1, 9)), int(rng.integers(1, 4000)), int(rng.integers(1, 3000))
    f = rng.random((b, c, n), dtype=np.float32); idx = rng.integers(0, n, (b, m)).astype(np.int32)
    ft = T(f).requires_grad_(True)
    out = gather_operation(ft, T(idx))
    go = rng.random((b, c, m), dtype=np.float32)
    (out * T(go)).sum().backward()
    check("gather", np.array_equal(out.detach().cpu().numpy(), oracle.gather_forward(f, idx))
          and np.allclose(ft.grad.cpu().numpy(), oracle.gather_backward(go, idx, n), rtol=1e-5, atol=1e-6), dict(b=b, c=c, n=n, m=m))
    # ---- p2i max backward (all radii, exact fixed point) and p2i sum forward / backward
    B, npts, C, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 1500)), int(rng.choice([1, 2])), int(rng.integers(4, 60)), int(rng.integers(4, 60))
    R = float(rng.choice([1.0, 3.0, 6.5, 12.0]))
    pts = ((rng.random((B * npts, 2)) * 1.2 - 0.1) * np.array([H - 1.0, W - 1.0])).astype(np.float32)
    feat = rng.random((B * npts, C)).astype(np.float32)
    bi = np.repeat(np.arange(B, dtype=np.int32), npts)
    bg = np.zeros((B, C, H, W), np.float32)
    og = rng.random((B, C, H, W)).astype(np.float32)
    o, ids = oracle.p2i_max_forward(pts, feat, bi, bg, R)
    gp0, gf0, gb0 = oracle.p2i_max_backward(og, ids, pts, feat, R)
    gp, gf, gb = ext.p2i_max_backward_multi_gpu(T(og)[None], T(ids)[None], T(pts), T(feat), 0, [R])
    so = ext.p2i_sum_forward_gpu(T(pts), T(feat), T(bi), T(bg), 0, R)
    sgp, sgf = ext.p2i_sum_backward_gpu(T(og), T(pts), T(feat), T(bi), 0, R)
    s0 = oracle.p2i_sum_forward(pts, feat, bi, bg, R)
    sgp0, sgf0 = oracle.p2i_sum_backward(og, pts, feat, bi, R)
    scale = max(1.0, float(np.abs(s0).max()))
    check("p2i_bwd", np.allclose(gp.cpu().numpy(), gp0, rtol=5e-5, atol=5e-6) and np.allclose(gf.cpu().numpy(), gf0, rtol=5e-5, atol=5e-6)
          and np.array_equal(gb.cpu().numpy(), gb0) and np.allclose(so.cpu().numpy(), s0, rtol=5e-5, atol=5e-6 * scale)
          and np.allclose(sgp.cpu().numpy(), sgp0, rtol=1e-4, atol=2e-5 * scale) and np.allclose(sgf.cpu().numpy(), sgf0, rtol=1e-4, atol=1e-5 * scale),
          dict(B=B, n=npts, C=C, H=H, W=W, R=R))
    # ---- gridding / reverse / cubic sampling
    scale_g = int(rng.choice([4, 8, 16])); b, n = int(rng.integers(1, 3)), int(rng.integers(1, 700))
    pc = ((rng.random((b, n, 3), dtype=np.float32) * 2 - 1) * 0.95).astype(np.float32)
    pt = T(pc * (scale_g // 2)).requires_grad_(True)
    grid = GriddingFunction.apply(scale_g // 2, pt)
    gg = rng.random(grid.shape, dtype=np.float32)
    (grid * T(gg)).sum().backward()
    og_, w, ix = oracle.gridding_forward(pc * (scale_g // 2), scale_g)
    check("gridding", np.allclose(grid.detach().cpu().numpy(), og_, rtol=1e-5, atol=1e-6)
          and np.allclose(pt.grad.cpu().numpy(), oracle.gridding_backward(gg, w, ix), rtol=1e-5, atol=1e-6), dict(scale=scale_g, b=b, n=n))
    rg = rng.random((b, scale_g, scale_g, scale_g), dtype=np.float32)
    rp = GriddingReverseFunction.apply(scale_g, T(rg))
    check("gridding_reverse", np.allclose(rp.cpu().numpy(), oracle.gridding_reverse_forward(rg.reshape(b, -1), scale_g), rtol=1e-5, atol=1e-6),
          dict(scale=scale_g, b=b))
    s = int(rng.choice([4, 8, 16])); c = int(rng.integers(1, 20)); ns = int(rng.choice([1, 2]))
    feat = rng.random((b, c, s, s, s), dtype=np.float32)
    ft = T(feat).requires_grad_(True)
    pv = (pc * (s / 2) + s / 2).astype(np.float32)
    out = CubicFeatureSamplingFunction.apply(T(pv), ft, ns)
    go = rng.random(out.shape, dtype=np.float32)
    (out * T(go)).sum().backward()
    oo, oidx = oracle.cubic_forward(pv, feat, ns)
    check("cubic", np.array_equal(out.detach().cpu().numpy(), oo)
          and np.allclose(ft.grad.cpu().numpy(), oracle.cubic_backward(go, oidx, c, s, ns), rtol=1e-5, atol=1e-6), dict(s=s, c=c, ns=ns, n=n))
    # ---- k-NN (both paths agree as sets on clearly separated rows) + edge features
    b, c, n, k = int(rng.integers(1, 3)), int(rng.choice([3, 8, 40, 130])), int(rng.integers(40, 700)), int(rng.choice([1, 4, 8]))
    xk = rng.standard_normal((b, c, n)).astype(np.float32)
    ia, ib = knn_fused(T(xk), k).cpu().numpy(), knn_unfused(T(xk), k).cpu().numpy()
    x64 = xk.astype(np.float64); bad = 0
    for bb in range(b):
        xx = (x64[bb] ** 2).sum(0); dmat = xx[:, None] + xx[None, :] - 2.0 * x64[bb].T @ x64[bb]
        srt = np.sort(dmat, axis=1); clear = (srt[:, k] - srt[:, k - 1]) > 2e-5 * (xx.max() * 3.0)
        exact = np.argsort(dmat, axis=1, kind="stable")[:, :k]
        for i in np.nonzero(clear)[0]:
            bad += (set(ia[bb, i]) != set(exact[i])) + (set(ib[bb, i]) != set(exact[i]))
    gf_ = get_graph_feature(T(xk), k=k, idx=T(ia)).cpu().numpy()
    check("knn", bad == 0 and np.array_equal(gf_, oracle.graph_feature(xk, ia)), dict(b=b, c=c, n=n, k=k, bad=bad))
print(f"fuzz3: {runs} rounds, mismatches: {fails if fails else 'none'}")
sys.exit(1 if fails else 0)
