"""Minimum density sampling: parity against the oracle (dense + surface regime, several batch sizes = team
geometries) and ms per call.  SN_MDS_G=1 turns the dense-regime teams off (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
def ms(fn, K=2):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K
if "--parity" in sys.argv:
    for b, n, m, mm in ((3, 19384, 4096, 0.05), (1, 19384, 2500, 0.08), (9, 5000, 3000, 0.06), (5, 19384, 3000, 0.0085),
                        (33, 3000, 2000, 0.07), (4, 2048, 2048, 0.2)):
        x = torch.rand(b, n, 3, generator=g)
        mml = torch.full((b,), mm) * (1 + 0.1 * torch.rand(b, generator=g))
        want = oracle.mds(x.numpy(), m, mml.numpy(), exp_mode=1)
        got = minimum_density_sample(x.to(dev), m, mml.to(dev)).cpu().numpy()
        print(f"parity b={b} n={n} m={m} mml={mm}: {bool(np.array_equal(got, want))}", flush=True)
x = torch.rand(32, 19384, 3, generator=g).to(dev)
for b in (32, 8, 4, 1):
    for mm in (0.0853, 0.05, 0.03, 0.0085):
        mml = torch.full((b,), mm, device=dev)
        print(f"mds B={b} n=19384 m=16384 mml={mm}: {ms(lambda: minimum_density_sample(x[:b].contiguous(), 16384, mml)):.2f} ms", flush=True)
