"""Run one EMD forward (C2) and print per-iteration unassigned counts (from the oracle-free
device path: we re-run with iters=1..k is too slow, so just read stats deltas)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g).to(dev); y = torch.rand(B, N, 3, generator=g).to(dev)
emd_forward_raw(x, y, 0.005, 50)
torch.cuda.synchronize()
prev = 0
out = []
for it in (1, 2, 3, 5, 10, 20, 30, 40, 50):
    st = torch.zeros(2, dtype=torch.int64, device=dev)
    emd_forward_raw(x, y, 0.005, it, st)
    torch.cuda.synchronize()
    out.append((it, st[0].item() / N / B))
print("cumulative mean unassigned per cloud:", out)
