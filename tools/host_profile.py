"""Where does the HOST spend its time while it enqueues a bench step?  cProfile over 200 steps (HO_B clouds)."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
pred, gt = bench.make_inputs(dev, 0, 1, "weak")
b = int(os.environ.get("HO_B", "4"))
pred, gt = pred[:b].contiguous(), gt[:b].contiguous()
hp = bench.HotPath(dev, [5.0, 7.0, 10.0])
for _ in range(10):
    hp.step_overlapped(pred, gt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    hp.step_overlapped(pred, gt)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:70]))
