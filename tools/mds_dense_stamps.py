"""Where a round of the sampler's dense-regime team kernel spends its time (wave 0 of member 0 of cloud 0).
Needs the stamps build:  tools/build_variant.sh mdsstamps mds.hip -DSN_MDS_STAMPS ;  AB_LIB=tools/ab/lib_mdsstamps.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparenet_amd._lib as _L
_L.LIB_PATH = os.path.abspath(os.environ.get("AB_LIB", "tools/ab/lib_mdsstamps.so"))
from sparenet_amd.cuda.MDS.MDS_module import minimum_density_sample
import sparenet_amd

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
x = torch.rand(32, 19384, 3, generator=g).to(dev)
lib = sparenet_amd.lib()
names = ["updates of the accepted picks", "arg-min + second density + barrier A", "store + poll of the team's words",
         "replay among the candidates", "barrier B + hand-over"]
for b in (32, 4):
    for mm in (0.0853, 0.05):
        mml = torch.full((b,), mm, device=dev)
        xs = x[:b].contiguous()
        minimum_density_sample(xs, 16384, mml); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        lib.sn_mds_debug_stamps(out, 1)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); minimum_density_sample(xs, 16384, mml); e.record(); torch.cuda.synchronize()
        lib.sn_mds_debug_stamps(out, 1)
        rounds, exch = max(1, out[5]), max(1, out[6])
        per = [out[i] * 10.0 / rounds for i in range(5)]   # ns per PICK (100 MHz ticks)
        print(f"mds dense B={b} mml={mm}: {a.elapsed_time(e):.2f} ms per call (stamps build), {rounds} picks in {exch} exchanges "
              f"({rounds / exch:.2f} picks per exchange), {sum(per):.0f} ns per pick = " + ", ".join(f"{n} {p:.0f}" for n, p in zip(names, per)), flush=True)
