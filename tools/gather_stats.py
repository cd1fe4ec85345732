"""Survivor statistics of the binned gather (needs the diag build: `make -C sparenet_amd/csrc diag`,
AB_LIB=tools/ab/lib_diag.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
from sparenet_amd.utils.p2i_utils import ComputeDepthMaps
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
data = (torch.rand(32, 16384, 3, generator=g) - 0.5).to(dev)
cdm = ComputeDepthMaps("orthorgonal", 1.0, 256).to(dev)
lib = _L.lib()
out = (ctypes.c_ulonglong * 8)()
for radii in ([5.0, 7.0, 10.0], [10.0], [5.0]):
    cdm(data, view_id=0, radius_list=radii); torch.cuda.synchronize()
    lib.sn_p2i_gather_diag(out, 1)
    cdm(data, view_id=0, radius_list=radii); torch.cuda.synchronize()
    lib.sn_p2i_gather_diag(out, 1)
    w, b, c, s, pairs, upd, amb, walk = [int(v) for v in out]
    rings, walk = walk >> 32, walk & 0xffffffff   # tiles that stopped at a ring boundary (round 6's ring-level test)
    print(f"radii {radii}: per 8x8 tile: batches {b / w:.1f}, candidates {c / w:.0f}, survive the cull {s / w:.1f}; "
          f"in-range (pixel, radius, candidate) pairs valued with the fp32 series {pairs / w:.0f}, of which inside the band "
          f"of the running best {upd / w:.0f}; pixel slots settled by winner + runner-up {amb / w:.3f}, by the exact walk "
          f"{walk / w:.5f} (of {64 * len(radii)} slots); tiles whose outer ring was skipped {rings / w:.3f}")
