"""Per-wave timeline of the last bid launch (needs the wall-clock diag build: AB_LIB=tools/ab/lib_diag.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g).to(dev); y = torch.rand(B, N, 3, generator=g).to(dev)
for it in (1, 10, 50):
    st = torch.zeros(8 + 1024 * 16 * 14, dtype=torch.int64, device=dev)
    emd_forward_raw(x, y, 0.005, it, st); torch.cuda.synchronize()
    r = st[8:].view(-1, 14).cpu().numpy().astype(np.float64)
    r = r[r[:, 0] > 0]
    r = r[r[:, 0] >= r[:, 0].max() - 100 * (300 if it > 2 else 3000)]
    pc = lambda a: " ".join(f"{np.percentile(a, q):7.1f}" for q in (10, 50, 90, 99, 100))
    print(f"iter {it}: waves {len(r)}  (p10 p50 p90 p99 max)")
    names = ["setup us", "scan us", "post us", "visited", "batches", "hit blocks", "hit us", "batch us", "worth us", "mfma",
             "t0", "queued", "level-2 pass"]
    print(f"   wave lifetimes us {pc((r[:, 0] - r[:, 11]) / 100)}")
    for i, name in enumerate(names):
        if name == "t0": continue
        a = r[:, i + 1] / (100 if name.endswith("us") else 1)
        print(f"   {name:10s}", pc(a))
