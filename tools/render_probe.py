import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.utils.p2i_utils import ComputeDepthMaps
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
data = (torch.rand(32, 16384, 3, generator=g) - 0.5).to(dev)
cdm = ComputeDepthMaps("orthorgonal", 1.0, 256).to(dev)
PER_VIEW = os.environ.get("PER_VIEW") == "1"   # the reference's view-by-view loop instead of one pass
def step():
    p = data.clone().requires_grad_(True)
    if PER_VIEW:
        acc = None
        for v in range(8):
            m = cdm(p, view_id=v, radius_list=[5.0, 7.0, 10.0]).mean()
            acc = m if acc is None else acc + m
    else:
        acc = cdm.forward_views(p, range(8), [5.0, 7.0, 10.0]).mean() * 8
    acc.backward()
for _ in range(2): step()
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): step()
e.record(); torch.cuda.synchronize()
print(f"render fwd+bwd 8 views x 3 radii: {s.elapsed_time(e)/3:.2f} ms")
