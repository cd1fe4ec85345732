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
print("iter: U_mean items cyc/item hitcyc/item enq_iters hits batches batchcyc/item rounds/batch")
for it in (1, 2, 3, 7, 12, 20, 30, 50):
    a = run(it - 1); b = run(it); dd = b - a; rec = dd[8:].view(-1, 8).sum(0).tolist(); d = dd[:8].tolist()
    w = max(rec[5], 1)
    print(f"{it:3d}: {d[0]/N/B:8.1f} {rec[5]:6d} {rec[0]/w:10.0f} {rec[1]/w:10.0f} {rec[2]/w:8.1f} {rec[3]/w:8.1f} {rec[4]/w:7.1f} {rec[6]/w:10.0f} {rec[7]/max(rec[4],1):6.2f}")
