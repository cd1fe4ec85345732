import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
from sparenet_amd import _lib
dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g).to(dev); y = torch.rand(B, N, 3, generator=g).to(dev)
lib = _lib.lib()
def run(tag, a, b, st):
    emd_forward_raw(a, b, 0.005, 50, st); torch.cuda.synchronize()
    lib.sn_prof_reset(); lib.sn_prof_enable(1)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        emd_forward_raw(a, b, 0.005, 50, st)
    e.record(); torch.cuda.synchronize(); lib.sn_prof_enable(0)
    ms = ctypes.c_double(0); n = lib.sn_prof_read(b"emd_bid", ctypes.byref(ms))
    print(f"{tag}: bid {ms.value/3:.2f} ms/call, whole call {s.elapsed_time(e)/3:.2f} ms")
run("no stats", x, y, None)
run("with stats", x, y, torch.zeros(2, dtype=torch.int64, device=dev))
run("swapped (y,x)", y, x, None)
g2 = torch.Generator().manual_seed(99)
x2 = torch.rand(B, N, 3, generator=g2).to(dev); y2 = torch.rand(B, N, 3, generator=g2).to(dev)
run("other seed", x2, y2, None)
