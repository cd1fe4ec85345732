"""One GPU's share of an N-GPU STRONG-scaling step (B = 32 / N clouds), timed on one MI355X: predicts the strong
curve the driver measures (the ops have no collective; the all-reduce of four floats is not included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hp = bench.HotPath(dev, [5.0, 7.0, 10.0])
g = torch.Generator().manual_seed(1234)
P = torch.rand(32, bench.N, 3, generator=g).to(dev)
G = torch.rand(32, bench.N, 3, generator=g).to(dev)
base = None
for n in (1, 2, 4, 8):
    b = 32 // n
    pred, gt = P[:b].contiguous(), G[:b].contiguous()
    best, table = hp.choose_schedule(pred, gt) or ("forced: " + hp.order, {})   # measured per batch size, as bench.py does
    for _ in range(5):
        hp.step_overlapped(pred, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        hp.step_overlapped(pred, gt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    base = base or ms
    timers = []
    for _ in range(10):
        hp.step(pred, gt, timers)
    torch.cuda.synchronize()
    seg = {}
    for i in range(1, len(timers)):
        name, ev = timers[i]
        if name != "start":
            seg[name] = seg.get(name, 0.0) + timers[i - 1][1].elapsed_time(ev) / 10
    print(f"N = {n}: {b:2d} clouds per GPU: {ms:.2f} ms per step -> speed-up {base / ms:.2f} of {n} ({base / ms / n:.0%})"
          f"   one stream: " + " ".join(f"{k} {v:.2f}" for k, v in seg.items()) + f" = {sum(seg.values()):.2f} ms"
          f"   schedule {best}: " + " ".join(f"{k} {v:.2f}" for k, v in table.items()))
