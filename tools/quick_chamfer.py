"""Scratch timing of chamfer fwd/bwd at C2 (B=32,N=16384) with HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparenet_amd.cuda.chamfer_distance import cd

dev = torch.device("cuda:0")
B, N = 32, 16384
x = torch.rand(B, N, 3, device=dev); y = torch.rand(B, N, 3, device=dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty_like(d1)
i1 = torch.empty(B, N, dtype=torch.int, device=dev); i2 = torch.empty_like(i1)
g1 = torch.empty_like(x); g2 = torch.empty_like(y)
gd = torch.rand(B, N, device=dev)
for _ in range(3):
    cd.forward_cuda(x, y, d1, d2, i1, i2)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
K = 20
s.record()
for _ in range(K):
    cd.forward_cuda(x, y, d1, d2, i1, i2)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / K
pairs = 2.0 * B * N * N
print(f"chamfer fwd {ms:.3f} ms  {pairs/ms/1e9:.2f} Tpairs/s  {pairs*9/ms/1e9:.1f} TFLOP/s(9 flop/pair)")
s.record()
for _ in range(K):
    cd.backward_cuda(x, y, g1, g2, gd, gd, i1, i2)
e.record(); torch.cuda.synchronize()
print(f"chamfer bwd {s.elapsed_time(e)/K*1000:.1f} us")
