"""Chamfer forward + backward on COLLAPSED predictions (every predicted point within 1e-3 of one point, as early in
training): the inverse neighbour lists are thousands of entries long.  ms per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.cuda.chamfer_distance import ChamferDistance
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
gt = torch.rand(32, 16384, 3, generator=g).to(dev)
for name, pred in (("uniform", torch.rand(32, 16384, 3, generator=g)), ("collapsed", 0.5 + 1e-3 * (torch.rand(32, 16384, 3, generator=g) - 0.5))):
    p = pred.to(dev).requires_grad_(True)
    def step():
        d1, d2 = ChamferDistance()(p, gt)
        (d1.mean() + d2.mean()).backward()
    step(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): step()
    b.record(); torch.cuda.synchronize()
    print(f"chamfer fwd+bwd B=32 N=16384, {name} predictions: {a.elapsed_time(b) / 3:.2f} ms")
