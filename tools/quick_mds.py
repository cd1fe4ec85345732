"""Scratch timing of MDS / gather / p2i at C2/C3 sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])  # A/B a saved build
from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample
from sparenet_amd.utils.p2i_utils import ComputeDepthMaps

dev = torch.device("cuda:0")
def timeit(fn, K=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K
x = torch.rand(32, 19384, 3, device=dev); mml = torch.full((32,), 0.0085, device=dev)
print(f"mds B=32 n=19384 m=16384: {timeit(lambda: minimum_density_sample(x, 16384, mml), K=3):.2f} ms")
for mm in (0.03, 0.1):
    mml2 = torch.full((32,), mm, device=dev)
    print(f"mds mml={mm}: {timeit(lambda: minimum_density_sample(x, 16384, mml2), K=2):.2f} ms")
if os.environ.get("MDS_ONLY"): sys.exit(0)
data = torch.rand(32, 16384, 3, device=dev) - 0.5
cdm = ComputeDepthMaps("orthorgonal", 1.0, 256).to(dev)
for radii in ([5.0, 7.0, 10.0], [0.02, 0.05], [10.0]):
    ms = timeit(lambda: [cdm(data, view_id=v, radius_list=radii) for v in range(8)], K=3)
    print(f"depthmaps 8 views radii={radii}: {ms:.2f} ms  -> {32*8*len(radii)/ms*1000:.0f} maps/s")
from sparenet_amd.cuda.p2i_op import ext
pos, feat = cdm.project(data, 0)
px = (pos + 1) / 2 * 255
bi = torch.arange(32, dtype=torch.int32, device=dev).repeat_interleave(16384)
bg = torch.zeros(32, 1, 256, 256, device=dev)
for R in (5.0, 10.0):
    ms = timeit(lambda: ext.p2i_max_forward_gpu(px, feat, bi, bg, 0, R), K=5)
    print(f"p2i max fwd only R={R}: {ms*1000:.0f} us")
ms = timeit(lambda: cdm.project(data, 0), K=5)
print(f"project glue only: {ms*1000:.0f} us")
