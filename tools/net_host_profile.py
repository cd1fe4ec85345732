"""Where does the HOST spend its time in one rank's share of BASELINE config 4 / 5?  (Round 4: rocprofv3 shows the GPU
busy for 148 of config 5's 397 ms per step -- the rest is the host.)  Prints the step time, then torch.profiler's CPU
table (self time per operator) and cProfile's view of the main thread for a few steady-state steps.
    python tools/net_host_profile.py config5 [trained_stand_in_damped|scattered_stand_in|random_init]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "config5"
state = sys.argv[2] if len(sys.argv) > 2 else "trained_stand_in_damped"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step = bench.make_network_step(dev, cfg, state)
for _ in range(3):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print(f"{cfg} {state}: host returns after / GPU done after (ms):", " ".join(f"{a:.0f}/{b:.0f}" for a, b in ts), flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(22)
    print(f"==== cProfile, 2 steps, by {key}\n" + "\n".join(l[:150] for l in s.getvalue().splitlines()[:40]), flush=True)
