"""EMD forward: wall time per call (HIP events) for a few batch sizes, and a parity check of the
current path against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

dev = torch.device("cuda:0")
N = 16384
g = torch.Generator().manual_seed(1234)
X = torch.rand(32, N, 3, generator=g); Y = torch.rand(32, N, 3, generator=g)
if os.environ.get("AB_DATA") == "surface":
    # what a training step sees (BASELINE configs 4-5): the ground truth on a surface (a sphere, patch ordered), the
    # prediction = the ground truth + 1 % noise.  Fewer unassigned bidders per iteration than on uniform cubes, but
    # contested targets: prices climb, and with them every reach
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    Y = bench.surface_like(32, N, g)
    X = (Y + float(os.environ.get("AB_NOISE", "0.01")) * torch.randn(32, N, 3, generator=g)).contiguous()
if os.environ.get("AB_DATA") == "scatter":
    # early training: the ground truth on a sphere, the prediction scattered around it (uniform offsets up to
    # AB_NOISE, default 0.3): every bidder is far from every target compared with the targets' spacing
    import bench
    Y = bench.surface_like(32, N, g)
    X = (Y + float(os.environ.get("AB_NOISE", "0.3")) * (2 * torch.rand(32, N, 3, generator=g) - 1)).contiguous()
mode = "persistent auction"
if "--parity" in sys.argv:
    for b, iters in ((3, 50), (1, 7), (9, 3)):
        x, y = X[:b].numpy(), Y[:b].numpy()
        d0, a0, aux = oracle.emd_forward(x, y, 0.005, iters, mt=True, return_aux=True)
        st = torch.zeros(2, dtype=torch.int64, device=dev)
        d, a = emd_forward_raw(X[:b].to(dev), Y[:b].to(dev), 0.005, iters, st)
        print(mode, "parity b", b, "iters", iters, bool(np.array_equal(a.cpu().numpy(), a0)),
              bool(np.array_equal(d.cpu().numpy(), d0)), int(st[0]) == aux["pairs_eff"], flush=True)
if "--parity32" in sys.argv:   # the benched batch (XCD-local teams), 50 iterations
    x, y = X.numpy(), Y.numpy()
    d0, a0, aux = oracle.emd_forward(x, y, 0.005, 50, mt=True, return_aux=True)
    for rep in range(3):
        st = torch.zeros(2, dtype=torch.int64, device=dev)
        d, a = emd_forward_raw(X.to(dev), Y.to(dev), 0.005, 50, st)
        print(mode, "parity b 32 iters 50 run", rep, bool(np.array_equal(a.cpu().numpy(), a0)),
              bool(np.array_equal(d.cpu().numpy(), d0)), int(st[0]) == aux["pairs_eff"], flush=True)
_BS = tuple(int(v) for v in os.environ["AB_BS"].split(",")) if os.environ.get("AB_BS") else None
for b in (_BS or ((32,) if os.environ.get("AB_QUICK") else (32, 4, 1))):
    x, y = X[:b].to(dev), Y[:b].to(dev)
    for iters in ((50,) if (os.environ.get("AB_QUICK") or _BS) else (1, 10, 50)):
        emd_forward_raw(x, y, 0.005, iters); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            emd_forward_raw(x, y, 0.005, iters)
        e1.record(); torch.cuda.synchronize()
        print(f"{mode}: B={b} iters={iters}: {e0.elapsed_time(e1)/5:.3f} ms per call", flush=True)

if os.environ.get("SN_EMD_DIAG"):
    for b in (int(os.environ.get("AB_DIAG_B", "32")),):   # AB_DIAG_B=4: the strong-scaling share of an 8-GPU job
        x, y = X[:b].to(dev), Y[:b].to(dev)
        _, _, ws = emd_forward_raw(x, y, 0.005, 50, return_workspace=True); torch.cuda.synchronize()
        off = _L.lib().sn_emd_diag_offset(b, N)
        v = ws[off:off + 8 * (16 + 64 * 64)].view(torch.int64).cpu().numpy()
        names = ["compact", "-", "bid", "bar1", "award", "bar2", "-", "-"]
        print("teams with every workgroup on one XCD (plain stores):", int(v[12]))
        print("team 0 / wg 0 phase time, us over the call:", {n_: round(float(v[4 + i]) / 100.0, 1) for i, n_ in enumerate(names)})
        if os.environ.get("SN_EMD_DIAG") == "2":
            t = v[16:16 + 50 * 64].reshape(50, 8, 8) / 100.0   # [it, wg, phase] us
            print("per-iteration phase time in us, mean over the team's 8 workgroups / max:")
            for it in (0, 1, 2, 3, 5, 8, 12, 20, 30, 40, 49):
                print(f"  it {it:2d}:", " ".join(f"{names[p]} {t[it,:,p].mean():5.1f}/{t[it,:,p].max():5.1f}" for p in range(8)))
            tail = t[10:]
            print("  tail mean per iteration:", {names[p]: round(float(tail[:, :, p].mean()), 1) for p in range(8)},
                  "sum", round(float(tail.mean(1).sum(1).mean()), 1))
        if os.environ.get("BID_STAMPS"):   # a library built with -DSN_BID_STAMPS (tools/build_variant.sh)
            w = v[16 + 3200:16 + 3200 + 3 * 16 * 16].reshape(3, 16, 16).astype(float)
            w = w[:, w.sum((0, 2)) > 0, :]      # waves that exist (8-wave workgroups leave the upper rows empty)
            names2 = ["setup", "boxes", "visits (filter)", "drain batches", "final+merge", "operand wait", "#visits", "#drains",
                      "#g-blocks", "#g-blocks hit", "#enqueue rounds", "#hits", "#exact evals", "#election rounds", "enqueue", "-"]
            per_it = float(os.environ.get("BID_STAMPS_ITS", "40"))   # stamped iterations (SN_STAMP_FROM .. SN_STAMP_TO of the build)
            print(f"bid_group, {per_it:.0f} stamped iterations, per wave and iteration; us or counts (mean over the waves of 3 workgroups / max wave):")
            for i, n_ in enumerate(names2):
                sc = 100.0 if (i < 6 or i == 14) else 1.0
                print(f"  {n_:16s} {w[:, :, i].mean() / sc / per_it:7.2f} / {w[:, :, i].max() / sc / per_it:7.2f}")
            tot = (w[:, :, :6].sum(2) + w[:, :, 14]) / 100.0 / per_it
            print("  total per wave: mean", round(float(tot.mean()), 2), "max", round(float(tot.max()), 2), "per-wg max", np.round(tot.max(1), 1))
