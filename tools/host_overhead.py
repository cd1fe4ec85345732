"""How long does the HOST need to enqueue one bench step (and its parts), compared with the GPU time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
pred, gt = bench.make_inputs(dev, 0, 1, "weak")
b = int(os.environ.get("HO_B", "32"))      # HO_B=4: the 8-GPU strong-scaling share
pred, gt = pred[:b].contiguous(), gt[:b].contiguous()
print(f"clouds per step: {b}")
hp = bench.HotPath(dev, [5.0, 7.0, 10.0])
for _ in range(5):
    hp.step_overlapped(pred, gt)
torch.cuda.synchronize()
for name, fn in (("render_all (8 views fwd+bwd)", lambda: hp._render_all(pred)),
                 ("distance losses", lambda: hp._distance_losses(pred, gt)),
                 ("whole overlapped step", lambda: hp.step_overlapped(pred, gt))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
    print(f"{name}: host enqueue {t_host * 1e3:.2f} ms per call, with the GPU drained {t_all * 1e3:.2f} ms")
