"""Per-iteration phase counters of the bid kernel (needs a -DSN_EMD_DIAG build: AB_LIB=tools/ab/lib_diag.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g).to(dev); y = torch.rand(B, N, 3, generator=g).to(dev)
def run(it):
    st = torch.zeros(8 + 1024 * 16 * 8, dtype=torch.int64, device=dev)
    emd_forward_raw(x, y, 0.005, it, st); torch.cuda.synchronize()
    return st.cpu()
prev = run(0)
print("iter: U_mean items setup scan finaldrain post (cycles/item)   kernel span (cycles, max end - min start)")
for it in (1, 2, 7, 20, 30, 50):
    a = run(it - 1); b = run(it); dd = b - a; recs = dd[8:].view(-1, 8); rec = recs.sum(0).tolist(); d = dd[:8].tolist()
    w = max(rec[5], 1)
    bb = b[8:].view(-1, 8); used = bb[:, 5] > 0
    # stamps are absolute (assigned, not accumulated): use the last run's values
    span = (bb[used][:, 7].max() - bb[used][:, 6].min()).item() if used.any() else 0
    print(f"{it:3d}: {d[0]/N/B:8.1f} {rec[5]:6d} {rec[0]/w:9.0f} {rec[1]/w:9.0f} {rec[2]/w:9.0f} {rec[3]/w:9.0f}   span {span}")
