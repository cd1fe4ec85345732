"""Scratch timing of EMD + expansion at C2 (B=32,N=16384)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])  # A/B a saved build
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
from sparenet_amd.cuda.expansion_penalty.expansion_penalty_module import expansionPenaltyFunction

dev = torch.device("cuda:0")
B, N = 32, 16384
x = torch.rand(B, N, 3, device=dev); y = torch.rand(B, N, 3, device=dev)
st = torch.zeros(2, dtype=torch.int64, device=dev)
for it in (1, 50):
    emd_forward_raw(x, y, 0.005, it)
    torch.cuda.synchronize()
    st.zero_()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    K = 3
    s.record()
    for _ in range(K):
        d, a = emd_forward_raw(x, y, 0.005, it, st)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / K
    pairs = st[0].item() / K
    print(f"emd iters={it}: {ms:.3f} ms  pairs_eff={pairs:.3e}  {pairs/ms/1e9:.3f} Tpairs/s  active_iters={st[1].item()/K}")
if os.environ.get("EMD_ONLY"): sys.exit(0)
for _ in range(2):
    expansionPenaltyFunction.apply(x, 512, 1.5)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    expansionPenaltyFunction.apply(x, 512, 1.5)
e.record(); torch.cuda.synchronize()
print(f"expansion fwd {s.elapsed_time(e)/10*1000:.1f} us")
