import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g).to(dev); y = torch.rand(B, N, 3, generator=g).to(dev)
prev = None
for it in (1, 2, 3, 5, 10, 20, 30, 50):
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    emd_forward_raw(x, y, 0.005, it, st); torch.cuda.synchronize()
    s = st.tolist()
    if prev is not None:
        dp, dg, de = s[0] - prev[1][0], s[2] - prev[1][2], s[3] - prev[1][3]
        nit = it - prev[0]
        print(f"iters {prev[0]+1}..{it}: pairs/it {dp/nit:.3e}  wave-steps/it {dp/nit/64:.3e}  slow groups/it {dg/nit:.3e}  exact evals/it {de/nit:.3e}  exact/wave-step {de/(dp/64):.3f}")
    else:
        print(f"iter 1: pairs {s[0]:.3e} wave-steps {s[0]/64:.3e} slow groups {s[2]:.3e} exact evals {s[3]:.3e} exact/wave-step {s[3]/(s[0]/64):.3f}")
    prev = (it, s)
