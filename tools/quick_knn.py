"""Scratch timing of the fused k-NN kernel against GEMM + ranking at the generator's EdgeConv sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparenet_amd.cuda.knn import knn_fused as knn, knn_unfused

dev = torch.device("cuda:0")
def timeit(fn, K=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K
g = torch.Generator().manual_seed(0)
for (b, c, n, k) in ((32, 3, 3000, 8), (32, 256, 3000, 8), (32, 512, 3000, 8), (32, 256, 3000, 20), (32, 64, 16384, 8)):
    x = torch.rand(b, c, n, generator=g).to(dev)
    tf = timeit(lambda: knn(x, k))
    tu = timeit(lambda: knn_unfused(x, k)) if n <= 4096 else float("nan")
    fl = 2.0 * b * n * n * c
    print(f"B={b} C={c} N={n} k={k}: fused {tf:.3f} ms ({fl / tf / 1e9:.1f} TFLOP/s)  gemm+rank {tu:.3f} ms")
