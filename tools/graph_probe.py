"""Which ops can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed?  Per op: capture one forward (+
backward), replay three times, compare every replay with the eager result, time a replay.  GP_B = clouds (default 4),
GP_OPS = comma-separated subset.  Findings on ROCm 7.2 / MI355X (round 3, before the refusals were added): the
expansion penalty replays bit-identically (0.4 ms); the Chamfer kernels' replay died with a memory access fault; the
whole step's replay took 5.4 s (the persistent auction running into its barrier time-out).  sn_emd_forward,
sn_chamfer_forward_sorted and sn_chamfer_backward therefore refuse to be captured (common.hpp, SN_REFUSE_CAPTURE) and
show up here as "capture failed"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
b = int(os.environ.get("GP_B", "4"))
pred, gt = bench.make_inputs(dev, 0, 1, "weak")
pred, gt = pred[:b].contiguous(), gt[:b].contiguous()
hp = bench.HotPath(dev, [5.0, 7.0, 10.0])
ops = {"expansion": lambda: hp._loss_expansion(pred), "chamfer": lambda: hp._loss_cd(pred, gt),
       "emd": lambda: hp._loss_emd(pred, gt), "render": lambda: hp._render_all(pred)}
only = os.environ.get("GP_OPS")
for name, fn in ops.items():
    if only and name not in only.split(","):
        continue
    for _ in range(3):
        eager = fn().detach().clone()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = fn()
    except Exception as e:
        print(f"{name}: capture failed: {type(e).__name__} {str(e)[:200]}", flush=True)
        continue
    res = []
    for r in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        res.append((bool(torch.equal(out.detach(), eager)), round((time.perf_counter() - t0) * 1e3, 3)))
    print(f"{name} (B = {b}): replays (equal to eager, ms): {res}", flush=True)
    del g
