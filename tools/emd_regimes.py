"""The auction on the four kinds of data a training run shows it (uniform cubes = the benchmark; prediction = ground
truth + 1 % noise = a trained generator; prediction scattered +-0.3 around the surface = early training; the refine
stages of an UNTRAINED generator = the first steps of every run), B = 32 and 4: ms per call, optional per-iteration
phase times (SN_EMD_DIAG=2) and an oracle check (--parity: whole clouds, 50 iterations).

    python tools/emd_regimes.py [--parity] [--dump DIR] [regime ...]
    AB_LIB=tools/ab/lib_x.so ... : another build of the library
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sparenet_amd._lib as _L

if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])
import bench
from sparenet_amd.cuda.emd.emd_module import emd_forward_raw

N = 16384
REGIMES = ("uniform", "surface", "scatter", "untrained")


def regime_clouds(name, b, dev, seed=1234):
    """(prediction, ground truth), [b, N, 3] each, on `dev`."""
    return bench.emd_regime_clouds(name, b, dev, seed)


def phases(x, y, b):
    _, _, ws = emd_forward_raw(x, y, 0.005, 50, return_workspace=True)
    torch.cuda.synchronize()
    off = _L.lib().sn_emd_diag_offset(b, N)
    v = ws[off:off + 8 * (16 + 64 * 64)].view(torch.int64).cpu().numpy()
    names = ["compact", "-", "bid", "bar1", "award", "bar2", "-", "-"]
    print("   team 0 / wg 0 phase time, us over the call:",
          {n_: round(float(v[4 + i]) / 100.0, 1) for i, n_ in enumerate(names) if n_ != "-"})
    if os.environ.get("SN_EMD_DIAG") == "2":
        t = v[16:16 + 50 * 64].reshape(50, 8, 8) / 100.0   # [it, wg, phase] us
        use = [0, 2, 3, 4, 5]
        for it in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 30, 40, 49):
            print(f"   it {it:2d}:", " ".join(f"{names[p]} {t[it,:,p].mean():6.1f}/{t[it,:,p].max():6.1f}" for p in use))
        print("   sum over iterations of mean-over-wgs:", {names[p]: round(float(t[:, :, p].mean(1).sum()), 1) for p in use})
        print("   sum over iterations of max-over-wgs :", {names[p]: round(float(t[:, :, p].max(1).sum()), 1) for p in use})


def main():
    dev = torch.device("cuda:0")
    dump = None
    argv = sys.argv[1:]
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = argv[i + 1]
        del argv[i:i + 2]
        os.makedirs(dump, exist_ok=True)
    todo = [a for a in argv if not a.startswith("--")] or REGIMES
    bs = tuple(int(v) for v in os.environ.get("AB_BS", "32,4").split(","))
    for name in todo:
        for b in bs:
            x, y = regime_clouds(name, b, dev)
            if dump and b == 4:
                np.savez_compressed(os.path.join(dump, f"emd_regime_{name}_b{b}.npz"), x=x.cpu().numpy(), y=y.cpu().numpy())
            for _ in range(2):
                emd_forward_raw(x, y, 0.005, 50)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                emd_forward_raw(x, y, 0.005, 50)
            e1.record()
            torch.cuda.synchronize()
            print(f"regime {name:9s} B={b:2d}: {e0.elapsed_time(e1) / reps:7.3f} ms per call", flush=True)
            if os.environ.get("SN_EMD_DIAG"):
                phases(x, y, b)
            if "--parity" in sys.argv and b == 4:
                import oracle
                t0 = time.time()
                d0, a0, aux = oracle.emd_forward(x.cpu().numpy(), y.cpu().numpy(), 0.005, 50, mt=True, return_aux=True)
                st = torch.zeros(2, dtype=torch.int64, device=dev)
                d, a = emd_forward_raw(x, y, 0.005, 50, st)
                print(f"   parity {name} B={b}: assignment {bool(np.array_equal(a.cpu().numpy(), a0))} dist "
                      f"{bool(np.array_equal(d.cpu().numpy(), d0))} pairs {int(st[0]) == aux['pairs_eff']} "
                      f"(oracle {time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
