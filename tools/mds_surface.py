"""Minimum density sampling 19384 -> 16384 on SURFACE-like clouds (a trained decoder's output + the partial input:
what the first sampler call of a real step sees; bench.py's other_ops cloud): ms per call and an index-exact check
against the oracle.  AB_LIB=tools/ab/lib_picksN.so compares builds (SN_MDS_PICKS = picks per round)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
import bench
from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
N, M = 16384, 3000
def cloud(b):
    gt = bench.surface_like(b, N, g)
    part = gt[:, torch.randperm(N, generator=g)[:M]] + 1e-3 * torch.randn(b, M, 3, generator=g)
    return torch.cat([gt + 0.01 * torch.randn(b, N, 3, generator=g), part], 1).contiguous()
def ms(fn, K=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K
x = cloud(32)
for mm in (0.010, 0.006, 0.02):
    for b in (32, 4):
        mml = torch.full((b,), mm) * (1 + 0.05 * torch.rand(b, generator=g))
        xd, md = x[:b].to(dev), mml.to(dev)
        t = ms(lambda: minimum_density_sample(xd, N, md))
        line = f"mds surface B={b} n={N + M} m={N} mml~{mm}: {t:.2f} ms"
        if "--parity" in sys.argv and b == 4:
            t0 = time.time()
            want = oracle.mds(x[:b].numpy(), N, mml.numpy(), exp_mode=1)
            got = minimum_density_sample(xd, N, md).cpu().numpy()
            line += f"   index-exact {bool(np.array_equal(got, want))} (oracle {time.time() - t0:.0f} s)"
        print(line, flush=True)
