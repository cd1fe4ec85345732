"""Randomised parity sweep, part 2 (by hand on a GPU box): minimum density sampling and the expansion
penalty bit-exact against the oracle, the multi-radius p2i splat within the suite's tolerances.
`python tools/fuzz_parity2.py [seconds] [seed]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample
from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyFunction
from sparenet_amd.cuda.p2i_op import ext

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def cloud(b, n, kind):
    if kind == "uniform":
        return rng.random((b, n, 3), dtype=np.float32)
    if kind == "lattice":
        return (rng.integers(0, 9, (b, n, 3)) / 8).astype(np.float32)
    if kind == "clustered":
        c = rng.random((b, 6, 3), dtype=np.float32)
        pick = np.take_along_axis(c, rng.integers(0, 6, (b, n, 1)).repeat(3, 2), 1)
        return (pick + 0.01 * rng.standard_normal((b, n, 3)).astype(np.float32)).astype(np.float32)
    if kind == "surface":
        v = rng.standard_normal((b, n, 3)).astype(np.float32)
        return (0.5 * v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32)
    if kind == "line":
        t = rng.random((b, n, 1), dtype=np.float32)
        return np.concatenate([t, 0.3 * t, -t], 2).astype(np.float32)
    raise ValueError(kind)


kinds = ["uniform", "lattice", "clustered", "surface", "line"]
t_end = time.time() + budget
counts = {"mds": [0, 0], "expansion": [0, 0], "p2i": [0, 0]}
while time.time() < t_end:
    # ---- MDS (index-exact)
    b = int(rng.integers(1, 4))
    n = int(rng.choice([300, 1024, 2500, 5000, 9000, 19384]))
    m = int(rng.integers(1, min(n, 1200) + 1))
    kind = rng.choice(kinds)
    x = cloud(b, n, kind)
    mml = (float(rng.choice([1e-4, 0.005, 0.01, 0.03, 0.1, 0.5])) * (1 + 0.2 * rng.random(b))).astype(np.float32)
    ref = oracle.mds(x, m, mml, exp_mode=1)
    got = minimum_density_sample(torch.from_numpy(x).to(dev), m, torch.from_numpy(mml).to(dev)).cpu().numpy()
    counts["mds"][0] += 1
    if not np.array_equal(got, ref):
        counts["mds"][1] += 1
        print("MDS MISMATCH", dict(b=b, n=n, m=m, kind=str(kind), mml=mml.tolist()), int((got != ref).sum()))
    # ---- expansion penalty (bit-exact)
    P = int(rng.choice([2, 8, 32, 64, 128, 256, 512]))
    npatch = int(rng.integers(1, 5))
    b = int(rng.integers(1, 4))
    kind = rng.choice(kinds)
    x = cloud(b, P * npatch, kind)
    if rng.random() < 0.3:   # ulp-level jitter on a lattice: near-ties of the rounded square roots
        x = (np.round(x * 8) / 8 + rng.integers(-3, 4, x.shape) * 2.0 ** -24).astype(np.float32)
    alpha = float(rng.choice([0.5, 1.0, 1.5, 2.5]))
    d0, a0, m0 = oracle.expansion_forward(x, P, alpha)
    d, a, mm = expansionPenaltyFunction.apply(torch.from_numpy(x).to(dev), P, alpha)
    counts["expansion"][0] += 1
    if not (np.array_equal(a.cpu().numpy(), a0) and np.array_equal(d.cpu().numpy(), d0)):
        counts["expansion"][1] += 1
        print("EXPANSION MISMATCH", dict(b=b, P=P, npatch=npatch, kind=str(kind), alpha=alpha))
    # ---- p2i max, several radii in one pass (suite tolerances: values 2e-6, ids 1e-4 of the pixels)
    B = int(rng.integers(1, 4)); npts = int(rng.choice([1, 50, 700, 3000])); C = int(rng.choice([1, 1, 2]))
    H = int(rng.integers(4, 90)); W = int(rng.integers(4, 90))
    radii = [float(r) for r in rng.choice([0.6, 1.0, 2.5, 5.0, 7.0, 10.0, 15.9, 18.0], int(rng.integers(1, 5)), replace=False)]
    pts = ((rng.random((B * npts, 2)) * 1.3 - 0.15) * np.array([H - 1.0, W - 1.0])).astype(np.float32)
    if rng.random() < 0.3:
        pts = np.round(pts)          # points on pixel centres: equal weights, ties by lowest id
    feat = (rng.random((B * npts, C)) - 0.2).astype(np.float32)
    bi = rng.integers(-1, B + 1, B * npts).astype(np.int32)
    bg = (rng.random((B, C, H, W)) * 0.1).astype(np.float32)
    out, ids = ext.p2i_max_forward_multi_gpu(*(torch.from_numpy(t).to(dev) for t in (pts, feat, bi, bg)), 0, radii)
    ok = True
    for r, R in enumerate(radii):
        o, i = oracle.p2i_max_forward(pts, feat, bi, bg, R)
        ok = ok and np.allclose(out[r].cpu().numpy(), o, rtol=2e-6, atol=1e-7) and (ids[r].cpu().numpy() != i).mean() < 1e-3
    counts["p2i"][0] += 1
    if not ok:
        counts["p2i"][1] += 1
        print("P2I MISMATCH", dict(B=B, n=npts, C=C, H=H, W=W, radii=radii))
print("fuzz2:", {k: f"{v[0]} runs, {v[1]} mismatches" for k, v in counts.items()})
sys.exit(1 if any(v[1] for v in counts.values()) else 0)
