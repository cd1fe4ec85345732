"""One rank's share of BASELINE config 4 / 5 (bench.make_network_step), a few steps: the command rocprofv3 wraps to
say where the step's time goes.   python tools/net_step.py config4|config5 [random_init|scattered_stand_in|trained_stand_in_damped] [steps]
NS_OVERLAP=0 turns the loss / sampler overlap off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "config4"
state = sys.argv[2] if len(sys.argv) > 2 else "trained_stand_in_damped"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step = bench.make_network_step(dev, cfg, state, batch_terms=os.environ.get("NS_BATCH_TERMS", "1") == "1")
for _ in range(int(os.environ.get("NS_WARMUP", "2"))):
    step()
torch.cuda.synchronize()
# the profiler's marker: one kernel that occurs nowhere else (tools/steady_stats.py counts from here on)
torch.bitwise_xor(torch.arange(4096, device=dev), 12345)
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{cfg} {state}: " + " ".join(f"{t:.1f}" for t in ts) + f" ms per step (median {sorted(ts)[len(ts) // 2]:.1f})", flush=True)
