import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])  # A/B a saved build
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw
from sparenet_amd import _lib
dev = torch.device("cuda:0")
B, N = 32, 16384
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g).to(dev); y = torch.rand(B, N, 3, generator=g).to(dev)
lib = _lib.lib()
for iters in (1, 50):
    emd_forward_raw(x, y, 0.005, iters); torch.cuda.synchronize()
    lib.sn_prof_reset(); lib.sn_prof_enable(1)
    for _ in range(3):
        emd_forward_raw(x, y, 0.005, iters)
    torch.cuda.synchronize(); lib.sn_prof_enable(0)
    ms = ctypes.c_double(0); n = lib.sn_prof_read(b"emd_auction", ctypes.byref(ms))
    print(f"iters={iters}: auction kernel {ms.value/3:.3f} ms per call ({n//3} launch per call)")
