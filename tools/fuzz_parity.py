"""Randomised parity sweep against the CPU oracle (test infrastructure, run by hand on a GPU box):
EMD and Chamfer on mixed geometries / sizes, bit-exact.  `python tools/fuzz_parity.py [seconds] [seed]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])  # A/B a saved build
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
from sparenet_amd.cuda.chamfer_distance.chamfer_distance import cd

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def cloud(b, n, kind):
    if kind == "uniform":
        return rng.random((b, n, 3), dtype=np.float32)
    if kind == "lattice":
        return (rng.integers(0, 7, (b, n, 3)) / 6).astype(np.float32)
    if kind == "clustered":
        c = rng.random((b, 5, 3), dtype=np.float32)
        pick = np.take_along_axis(c, rng.integers(0, 5, (b, n, 1)).repeat(3, 2), 1)
        return np.clip(pick + 0.004 * rng.standard_normal((b, n, 3)).astype(np.float32), 0, 1).astype(np.float32)
    if kind == "surface":
        v = rng.standard_normal((b, n, 3)).astype(np.float32)
        return (0.5 + 0.45 * v / np.linalg.norm(v, axis=2, keepdims=True)).astype(np.float32)
    if kind == "line":
        t = rng.random((b, n, 1), dtype=np.float32)
        return np.concatenate([t, 0.5 * t, 1.0 - t], 2).astype(np.float32)
    if kind == "far":
        return (rng.random((b, n, 3), dtype=np.float32) + 30.0).astype(np.float32)
    if kind == "dup":
        base = rng.random((b, max(n // 8, 1), 3), dtype=np.float32)
        return np.tile(base, (1, 8, 1))[:, :n].copy()
    raise ValueError(kind)


kinds = ["uniform", "lattice", "clustered", "surface", "line", "far", "dup"]
t_end = time.time() + budget
runs = fails = 0
while time.time() < t_end:
    # mostly small batches (one XCD = 32 workgroups per cloud up to 8 clouds, then teams of 16); one run in three a
    # batch of 32+ clouds: teams of 8 / 4 / 2 workgroups, teams that serve several clouds in turn
    big = rng.random() < 0.33
    b = int(rng.choice([32, 33, 48, 64, 70])) if big else int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17]))
    n = int(rng.choice([1024, 2048])) if big else int(rng.choice([1024, 2048, 3072, 4096]))
    k1, k2 = rng.choice(kinds), rng.choice(kinds)
    x, y = cloud(b, n, k1), cloud(b, n, k2)
    if "far" in (k1, k2):      # keep both clouds in one place: the auction needs 3 - dist - price > 0
        x, y = cloud(b, n, "far"), cloud(b, n, "far")
    eps = float(os.environ['FUZZ_EPS']) if os.environ.get('FUZZ_EPS') else float(rng.choice([0.005, 0.002, 0.01, -0.002]))
    iters = int(rng.choice([1, 2, 5, 17, 50]))
    d0, a0 = oracle.emd_forward(x, y, eps, iters, mt=True)
    d, a = emd_forward_raw(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), eps, iters)
    ok = np.array_equal(a.cpu().numpy(), a0) and np.array_equal(d.cpu().numpy(), d0)
    if not ok:
        bad_a = int((a.cpu().numpy() != a0).sum())
        print("  EMD mismatch:", bad_a, "assignments differ of", a0.size, dict(eps=eps, iters=iters, k1=str(k1), k2=str(k2), b=b, n=n))
    emd_ok = ok
    # Chamfer with ragged sizes through both search paths
    m = int(rng.choice([n, 1500, 2049, 5000]))
    z = cloud(b, m, rng.choice(kinds[:5]))
    r0 = oracle.chamfer_forward(x, z, mt=True)
    xt, zt = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    for fwd in (cd.forward_cuda, cd.forward_sorted_cuda):
        d1 = torch.empty(b, n, device=dev); d2 = torch.empty(b, m, device=dev)
        i1 = torch.empty(b, n, dtype=torch.int32, device=dev); i2 = torch.empty(b, m, dtype=torch.int32, device=dev)
        fwd(xt, zt, d1, d2, i1, i2)
        got = [t.cpu().numpy() for t in (d1, d2, i1, i2)]
        cd_ok = all(np.array_equal(g, r) for g, r in zip(got, r0))
        if not cd_ok:
            print("  Chamfer mismatch:", fwd.__name__, dict(b=b, n=n, m=m))
        ok = ok and cd_ok
    runs += 1
    if not ok:
        fails += 1
        print("MISMATCH", dict(b=b, n=n, m=m, k1=k1, k2=k2, eps=eps, iters=iters))
print(f"fuzz: {runs} configurations, {fails} mismatches")
sys.exit(1 if fails else 0)
