"""EMD forward: wall time per call (HIP events) for a few batch sizes, and a parity check of the
current path against the oracle.  SN_EMD_LAUNCHES=1 selects the launch-per-phase form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

dev = torch.device("cuda:0")
N = 16384
g = torch.Generator().manual_seed(1234)
X = torch.rand(32, N, 3, generator=g); Y = torch.rand(32, N, 3, generator=g)
mode = "per-phase launches" if os.environ.get("SN_EMD_LAUNCHES") == "1" else "persistent"
if "--parity" in sys.argv:
    for b, iters in ((3, 50), (1, 7), (9, 3)):
        x, y = X[:b].numpy(), Y[:b].numpy()
        d0, a0, aux = oracle.emd_forward(x, y, 0.005, iters, mt=True, return_aux=True)
        st = torch.zeros(2, dtype=torch.int64, device=dev)
        d, a = emd_forward_raw(X[:b].to(dev), Y[:b].to(dev), 0.005, iters, st)
        print(mode, "parity b", b, "iters", iters, bool(np.array_equal(a.cpu().numpy(), a0)),
              bool(np.array_equal(d.cpu().numpy(), d0)), int(st[0]) == aux["pairs_eff"], flush=True)
for b in (32, 4, 1):
    x, y = X[:b].to(dev), Y[:b].to(dev)
    for iters in (1, 10, 50):
        emd_forward_raw(x, y, 0.005, iters); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            emd_forward_raw(x, y, 0.005, iters)
        e1.record(); torch.cuda.synchronize()
        print(f"{mode}: B={b} iters={iters}: {e0.elapsed_time(e1)/5:.3f} ms per call", flush=True)
